/*
 * frieda_hip.h — C ABI of the MI355X-native FRI-DAS commit / prove path (libfrieda_hip.so).
 *
 * This is the drop-in boundary: the entry points a Rust `extern "C"` block in a frieda fork (Level A) or
 * an stwo `HipBackend` (Level B) would bind.  Plain pointers and sizes only; no C++/torch types; no
 * exceptions or unwinding cross it.  Every function returns an int status (FRIEDA_OK == 0).  Where the
 * reference panics (assert!/unwrap) the status is FRIEDA_ERR_INVARIANT and a Rust shim re-raises with
 * panic!; verifier rejections are reported through *ok, exactly like the reference's `bool`.
 *
 * Reference interfaces replaced (paths under /root/reference):
 *   Level A  src/lib.rs:31-43  api::{commit, generate_proof, verify}
 *            src/proof.rs:32-36 commit_and_generate_proof; src/proof.rs:19-26 struct Proof
 *   Level B  the stwo backend traits frieda instantiates with `CpuBackend`
 *            (src/commit.rs:15,17; src/proof.rs:47,48,52,58; src/utils.rs:21,28):
 *            PolyOps::{precompute_twiddles, evaluate}, MerkleOps::commit_on_layer,
 *            FriOps::{fold_circle_into_line, fold_line}, GrindOps::grind, plus the codec of
 *            src/utils.rs:10-33.
 *
 * Threading: a frieda_ctx owns one device, one stream, its twiddle cache and workspace.  A ctx is not
 * thread-safe; distinct ctxs are independent (one per GPU / host thread).  No global mutable state.
 */
#ifndef FRIEDA_HIP_H
#define FRIEDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRIEDA_OK 0
#define FRIEDA_ERR_ARG 1       /* null pointer / size out of the supported range */
#define FRIEDA_ERR_HIP 2       /* a HIP runtime call failed (see frieda_last_error) */
#define FRIEDA_ERR_INVARIANT 3 /* the reference would panic here */
#define FRIEDA_ERR_NOMEM 4
#define FRIEDA_ERR_FORMAT 5    /* malformed proof image */

#define FRIEDA_ABI_VERSION 1
#define FRIEDA_MAX_LOG_DOMAIN 28
#define FRIEDA_MAX_LOG_CELLS 12 /* reconstruction from up to 2^12 scattered cells */

typedef struct frieda_ctx frieda_ctx;
typedef struct frieda_proof frieda_proof;

/* stwo PcsConfig { pow_bits, fri_config: FriConfig { log_blowup_factor, log_last_layer_degree_bound,
 * n_queries } } as passed at src/lib.rs:36, src/proof.rs:109-116, benches/proof.rs:5-12 */
typedef struct {
    uint32_t pow_bits;
    uint32_t log_blowup_factor;
    uint32_t log_last_layer_degree_bound;
    uint32_t n_queries;
} frieda_pcs_config;

uint32_t frieda_abi_version(void);
const char* frieda_status_string(int status);
/* detail of the last failure on this ctx (valid until the next call on it) */
const char* frieda_last_error(const frieda_ctx* ctx);
/* What context creation had to degrade on this device (an LDS opt-in refused ...), one line each; "" when nothing was.  A
 * successful frieda_ctx_create leaves frieda_last_error empty: notes are not errors. */
const char* frieda_ctx_notes(const frieda_ctx* ctx);

/* ---- context -------------------------------------------------------------------------------- */
/* stream: a hipStream_t to run on (e.g. torch's current stream), or NULL to create a private one */
int frieda_ctx_create(int device, void* stream, frieda_ctx** out);
int frieda_ctx_destroy(frieda_ctx* ctx);
int frieda_ctx_synchronize(frieda_ctx* ctx);
/* twiddle policy: 1 (default) keep the per-domain twiddle tables in the ctx across calls; 0 regenerate
 * them on every call as the reference does (src/commit.rs:15) */
int frieda_ctx_set_twiddle_cache(frieda_ctx* ctx, int enabled);

/* The ctx keeps its device workspace (sized by the largest call so far: a few hundred bytes per blob byte, times the batch),
 * its pinned staging block and its twiddle tables between calls.  This frees them; the next call allocates what it needs. */
int frieda_ctx_release_workspace(frieda_ctx* ctx);

/* transcript policy: 0 (default) the Fiat-Shamir channel of generate_proof runs on the device inside the commit-phase
 * kernels; 1 evaluates it on the host between layers (one 32-byte D2H + synchronise per layer).  Proofs are identical;
 * configurations whose last FRI layer exceeds 2^11 points always use the host policy. */
int frieda_ctx_set_host_channel(frieda_ctx* ctx, int enabled);

/* Tuning and A/B options of THIS context (none is needed in normal use; every option selects another kernel or plan for the SAME
 * result — DESIGN.md §10 lists them).  `name` is the name of the environment variable that sets the option's default when a context
 * is created, e.g. "FRIEDA_TAIL_RUN_LOG", "FRIEDA_NO_ENCODE_TREE_FUSION", "FRIEDA_HOST_DECOMMIT"; a context never re-reads the
 * environment after creation and no option is process-wide.  FRIEDA_ERR_ARG: unknown name or value out of range. */
int frieda_ctx_set_option(frieda_ctx* ctx, const char* name, int64_t value);

/* measurement aid: when enabled, HIP events are recorded on the ctx stream around every kernel launch.
 * frieda_ctx_kernel_timing_report synchronises the stream and writes a JSON object
 * {"kernels": [{"name", "launches", "total_ms", "alg_bytes"}]} (alg_bytes = algorithmic HBM bytes by the byte model
 * of DESIGN.md §5); returns the bytes needed including the NUL; reset != 0 clears the accumulated spans. */
int frieda_ctx_set_kernel_timing(frieda_ctx* ctx, int enabled);
/* host wall-clock marks of the last generate_proof on this ctx, ms since entry: [0] everything enqueued, [1] device work
 * complete (the one synchronisation), [2] queries drawn, [3] openings gathered ([2] = [3] = [1] unless the host fallback
 * planned the openings), [4] proof assembled, [5] host set-up before the first launch */
int frieda_ctx_last_prove_phases(const frieda_ctx* ctx, double out_ms[8]);
size_t frieda_ctx_kernel_timing_report(frieda_ctx* ctx, char* buf, size_t cap, int reset);
/* measurement aid: the pure-compute rate of the Merkle compression on THIS device, now — compressions per second with every
 * lane chaining Blake2s compressions on register-resident data (8 workgroups per CU), leaf-shaped (4 message words, 12 zero) and
 * node-shaped (16 words).  The path is bound by this rate, which differs by a few per cent between devices; bench.py quotes it
 * beside the measured kernels (roofline_valu) instead of a constant.  ~5 ms; synchronises the stream. */
int frieda_ctx_blake2s_ceiling(frieda_ctx* ctx, double* leaf_per_s, double* node_per_s);
/* the same after ~0.25 s of that load per shape (~0.5 s in all), with the clock the chip holds under it read inside the kernel (shader-clock counter against
 * the 100 MHz wall-clock counter, median over the workgroups): out = {leaf compressions/s, node compressions/s, leaf clock GHz,
 * leaf SIMD cycles per wave-compression, node clock GHz, node SIMD cycles per wave-compression}.  On MI355X the clock stays at
 * 2.3 - 2.4 GHz under this load and a compression costs ~2250 (leaf) / ~2400 (node) cycles per wave in the library's throughput
 * form (runs of one VALU rate class, the wave's priority raised for its slow-class runs; ~3950 for the scheduler's own fine
 * interleave, ~3150 / ~3300 with idle issue states instead of priorities: profiles/r05_blake2s_prio.txt, r05_blake2s_idle_sweep.txt):
 * the ceiling is the instruction stream and how the waves of a SIMD share the vector pipe. */
int frieda_ctx_blake2s_ceiling_ex(frieda_ctx* ctx, double out[6]);
/* diagnostic: the Fiat-Shamir transcript of the last finished generate_proof on this ctx (blob 0 of a batch) — what
 * FriProver::commit derives between src/proof.rs:52 and :58 and the Proof does not carry: per FRI layer (first, then inner)
 * the folding alpha drawn after its root (4 u32 each; the roots themselves are the layer commitments of the Proof), and the
 * channel digest the proof of work was keyed by (after mix_felts(last_layer_poly)).  *n_layers receives the layer count;
 * at most cap_layers alphas are written.  Exists so that a cargo-side dump of the reference's transcript can be diffed
 * against this path value by value (tools/dump_trace.py, tests/golden/trace_selfcheck.json). */
int frieda_ctx_last_transcript(const frieda_ctx* ctx, uint32_t* n_layers, uint32_t* alphas, size_t cap_layers,
                               uint8_t digest_before_grind[32]);

/* ---- Level A: frieda's public API ------------------------------------------------------------- */
/* api::commit(data, log_blowup_factor) -> [u8; 32]   (src/lib.rs:31, src/commit.rs:11-22) */
int frieda_commit(frieda_ctx* ctx, const uint8_t* data, size_t len, uint32_t log_blowup_factor, uint8_t out_root[32]);
/* same with the blob already resident in device memory; the root is written to device memory
 * (d_out_root, 32 B) asynchronously on the ctx stream — no host synchronisation */
int frieda_commit_device(frieda_ctx* ctx, const void* d_data, size_t len, uint32_t log_blowup_factor, void* d_out_root);

/* proof::commit_and_generate_proof(data, seed, cfg) -> (Commitment, Proof)   (src/proof.rs:32-77);
 * seed == NULL is Option::None */
int frieda_commit_and_generate_proof(frieda_ctx* ctx, const uint8_t* data, size_t len, const uint64_t* seed,
                                     frieda_pcs_config cfg, uint8_t out_commitment[32], frieda_proof** out);
int frieda_commit_and_generate_proof_device(frieda_ctx* ctx, const void* d_data, size_t len, const uint64_t* seed,
                                            frieda_pcs_config cfg, uint8_t out_commitment[32], frieda_proof** out);
/* The same, split in two so that several proofs can overlap on one GPU: _begin enqueues everything — encode, FRI commit
 * phase, proof of work, query sampling and openings — on the ctx stream and returns without synchronising; _finish waits
 * for it (once) and builds the proof from the openings the device left in pinned memory.  At most one proof (or batch) in
 * flight per ctx — use one ctx (= one stream + workspace) per in-flight proof.  While one is in flight, every other entry
 * point that sizes or writes the ctx's workspace or pinned block returns FRIEDA_ERR_ARG ("a proof is in flight on this
 * context") and leaves the proof untouched: frieda_commit*, frieda_commit_batch*, a second _begin, frieda_merkle_root,
 * frieda_merkle_commit_layer, frieda_grind, frieda_reconstruct*_device, frieda_circle_interpolate_cells,
 * frieda_ctx_release_workspace.  Level B calls that only read the twiddle cache and caller buffers stay available.
 * A blob passed to _begin_device must stay valid until _finish returns. */
int frieda_prove_begin(frieda_ctx* ctx, const uint8_t* data, size_t len, const uint64_t* seed, frieda_pcs_config cfg);
int frieda_prove_begin_device(frieda_ctx* ctx, const void* d_data, size_t len, const uint64_t* seed, frieda_pcs_config cfg);
int frieda_prove_finish(frieda_ctx* ctx, uint8_t out_commitment[32], frieda_proof** out);
/* Batches of small blobs (the reference's own bench sizes, benches/commit.rs:6-10, benches/proof.rs:14-21,
 * are 1 KiB .. 64 KiB blobs, where one proof is a chain of ~50 dependent launches on an almost idle chip).  `count` blobs of
 * `len` bytes each, blob i at data + i * stride (stride >= len); seeds == NULL is None for all, else seeds[i] is Some for blob i.
 * Every kernel of the path handles the whole batch in one launch, so the launch and Fiat-Shamir latency is paid once per batch;
 * results are exactly those of `count` separate calls.  out_commitments / out_roots: count * 32 bytes (host);
 * out_proofs: count handles, each released with frieda_proof_free.  Proof batches need the last FRI layer to have <= 2^11
 * points (log_last_layer_degree_bound + log_blowup_factor <= 11) and the default (device) transcript policy. */
int frieda_commit_and_generate_proof_batch(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count,
                                           const uint64_t* seeds, frieda_pcs_config cfg, uint8_t* out_commitments,
                                           frieda_proof** out_proofs);
int frieda_commit_and_generate_proof_batch_device(frieda_ctx* ctx, const void* d_data, size_t stride, size_t len, uint32_t count,
                                                  const uint64_t* seeds, frieda_pcs_config cfg, uint8_t* out_commitments,
                                                  frieda_proof** out_proofs);
/* split form, as frieda_prove_begin / _finish: _begin enqueues the device work of the whole batch and returns; _finish (same
 * count) waits and builds the proofs.  Two contexts alternating _begin / _finish keep the host-side assembly of one batch
 * under the device work of the next. */
int frieda_prove_batch_begin(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count, const uint64_t* seeds,
                             frieda_pcs_config cfg);
int frieda_prove_batch_begin_device(frieda_ctx* ctx, const void* d_data, size_t stride, size_t len, uint32_t count,
                                    const uint64_t* seeds, frieda_pcs_config cfg);
int frieda_prove_batch_finish(frieda_ctx* ctx, uint32_t count, uint8_t* out_commitments, frieda_proof** out_proofs);
int frieda_commit_batch(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count, uint32_t log_blowup_factor,
                        uint8_t* out_roots);
int frieda_commit_batch_device(frieda_ctx* ctx, const void* d_data, size_t stride, size_t len, uint32_t count,
                               uint32_t log_blowup_factor, uint8_t* out_roots);
/* api::generate_proof (src/lib.rs:36) */
int frieda_generate_proof(frieda_ctx* ctx, const uint8_t* data, size_t len, const uint64_t* seed, frieda_pcs_config cfg,
                          frieda_proof** out);
/* api::verify(proof, seed) -> bool  (src/lib.rs:41, src/proof.rs:79-101).  Host-only (the reference's
 * verifier is O(n_queries * log N) hashes).  *ok receives the bool; FRIEDA_ERR_INVARIANT where the
 * reference panics (src/proof.rs:166-173). */
int frieda_verify(const frieda_proof* proof, const uint64_t* seed, int* ok);
/* The sampling client's half of the README's flow (/root/reference/README.md:56-69; in src/ a sample IS a proof: the seed of
 * generate_proof decides, through the transcript, which positions are opened, src/proof.rs:40-42,60-66): verify as above and, when the
 * proof is accepted, also return WHERE it sampled — out_positions[i] (ascending, distinct) is the position in the bit-reversed
 * codeword of length 2^(log_size_bound + log_blowup_factor) whose four column values are evaluations[i]
 * (frieda_proof_evaluations).  Verified (position, value) pairs pooled from proofs with different seeds are exactly the input of
 * frieda_circle_interpolate_points / frieda_reconstruct_points_device (log_cell 0, cell r = evaluations[r]).  *n_positions receives
 * the count (0 when the proof is rejected); FRIEDA_ERR_ARG when cap is smaller than that. */
int frieda_verify_samples(const frieda_proof* proof, const uint64_t* seed, int* ok, uint32_t* out_positions, size_t cap, size_t* n_positions);

/* ---- batch policy: how a stream of equal-length blobs is cut into batched calls ("bytes in flight") -------------------------
 * Every kernel of a batched call covers all its blobs, so the launch / Fiat-Shamir latency chain is paid once per call: small
 * blobs want many per call, large blobs few (the device workspace grows with the count).  The library's rule — what
 * frieda_prove_many / frieda_commit_many apply per device, offered here to callers that drive _begin / _finish themselves (the
 * caller loop of benches/proof.rs:30-44 over many blobs):
 *     per_call = clamp(budget / frieda_workspace_bytes(blob), 1, min(4096, ceil(count / (calls_per_ctx * in_flight))))
 *     calls    = the smallest multiple of in_flight with no call above per_call, sizes equal to within one
 * budget: context option "FRIEDA_BATCH_BUDGET_MB" (0 = default: the workspace of sixteen proofs on a 2^24 domain, ~43 GB per call in flight — lower it on a shared device);
 * calls_per_ctx: option "FRIEDA_BATCH_CALLS_PER_CTX" (default 1: one call per context when the budget allows — measured equal to
 * two at the 2^22 / 2^24 domains and faster below, where a call is mostly its latency chain; raise it to bound a call's latency).
 * in_flight = the contexts taking turns (frieda_prove_many uses 2 per device).
 * frieda_workspace_bytes: device workspace one blob of `len` bytes adds to a batched call (prove != 0: commit_and_generate_proof,
 * else commit); 0 when the shape is outside what the entry points accept.
 * frieda_batch_plan: out_calls[0 .. *n_calls) receive the blob count of each call, in order (sum = count); ctx == NULL uses the
 * default options; out_calls == NULL only counts; FRIEDA_ERR_ARG when cap is too small or the shape is invalid. */
size_t frieda_workspace_bytes(size_t len, uint32_t log_blowup_factor, uint32_t log_last_layer_degree_bound, int prove);
int frieda_batch_plan(const frieda_ctx* ctx, size_t len, uint32_t log_blowup_factor, uint32_t log_last_layer_degree_bound, int prove,
                      uint32_t count, uint32_t in_flight, uint32_t* out_calls, size_t cap, uint32_t* n_calls);

/* ---- multi-GPU: a batch of independent blobs across the GPUs of one node ---------------------------------------------
 * What a caller looping api::commit / commit_and_generate_proof over blobs gets on an 8 x MI355X node
 * (src/lib.rs:31-38; benches/commit.rs:11-15, benches/proof.rs:30-44).  The path shards at blob granularity: blob i runs
 * on devices[i mod n], one host thread and two contexts per device (two calls in flight; a run of equal-length blobs on a
 * device goes through the batched kernels, cut into calls by the batch policy above — options set on frieda_multi_ctx(m, d)
 * apply to device slot d), no data-path collective.  The
 * only exchange is the gather of the 32-byte commitment roots: one ncclAllGather per device on a single-process
 * communicator (ncclCommInitAll; RCCL over xGMI), after which every device holds every root; the host reads device 0's
 * copy.  n == 1 needs no exchange and does not touch RCCL.  RCCL is bound at frieda_multi_create by dlopen("librccl.so.1")
 * (an instance already in the process — PyTorch's — is shared); creation fails with FRIEDA_ERR_HIP if it cannot be loaded.
 * blobs / lens: host arrays of `count` pointers / byte lengths (lengths may differ); out_roots / out_commitments:
 * count * 32 bytes in blob order; out_proofs: count handles, each released with frieda_proof_free; seeds == NULL is None
 * for every blob, else seeds[i] is Some.  A handle is not thread-safe; results are exactly those of `count` separate
 * single-GPU calls. */
typedef struct frieda_multi frieda_multi;
int frieda_multi_create(const int* devices, uint32_t n_devices, frieda_multi** out);
int frieda_multi_destroy(frieda_multi* m);
uint32_t frieda_multi_device_count(const frieda_multi* m);
const char* frieda_multi_last_error(const frieda_multi* m);
/* 1 when root gathers go through RCCL (n > 1, or FRIEDA_MULTI_FORCE_RCCL=1); collectives issued so far */
int frieda_multi_uses_rccl(const frieda_multi* m);
uint64_t frieda_multi_gather_count(const frieda_multi* m);
/* Each device's worker thread is pinned to the CPUs of that GPU's NUMA node (sysfs numa_node of its PCI function, intersected with
 * the process's affinity; FRIEDA_MULTI_NO_NUMA_PIN=1 or a platform that reports none: not pinned).  Returns how many CPUs slot d's
 * worker is pinned to (0: not pinned) and writes up to cap of them. */
uint32_t frieda_multi_near_cpus(const frieda_multi* m, uint32_t device_slot, int* out_cpus, size_t cap);
/* Frees the device workspaces (two per device, up to the batch budget each) and upload rings the handle keeps between calls;
 * twiddle caches stay.  The next call allocates again.  FRIEDA_ERR_ARG while a call is in flight. */
int frieda_multi_release_workspace(frieda_multi* m);
/* the first context of device slot d (e.g. to set a policy on it); owned by the handle */
frieda_ctx* frieda_multi_ctx(frieda_multi* m, uint32_t device_slot);
int frieda_commit_many(frieda_multi* m, const uint8_t* const* blobs, const size_t* lens, uint32_t count, uint32_t log_blowup_factor,
                       uint8_t* out_roots);
int frieda_prove_many(frieda_multi* m, const uint8_t* const* blobs, const size_t* lens, uint32_t count, const uint64_t* seeds,
                      frieda_pcs_config cfg, uint8_t* out_commitments, frieda_proof** out_proofs);

/* ---- struct Proof (src/proof.rs:19-26) accessors; fields are `pub` upstream, hence the setters --- */
void frieda_proof_free(frieda_proof* p);
int frieda_proof_clone(const frieda_proof* p, frieda_proof** out);
uint64_t frieda_proof_proof_of_work(const frieda_proof* p);
void frieda_proof_set_proof_of_work(frieda_proof* p, uint64_t nonce);
frieda_pcs_config frieda_proof_pcs_config(const frieda_proof* p);
uint32_t frieda_proof_log_size_bound(const frieda_proof* p);
/* evaluations: Vec<QM31>, 4 little-endian u32 coordinates each; pointer stays valid until the next
 * resize/free */
size_t frieda_proof_n_evaluations(const frieda_proof* p);
uint32_t* frieda_proof_evaluations(frieda_proof* p);
int frieda_proof_resize_evaluations(frieda_proof* p, size_t n);
/* FriProof: layer 0 = first_layer, 1..n_inner = inner_layers */
size_t frieda_proof_n_inner_layers(const frieda_proof* p);
const uint8_t* frieda_proof_layer_commitment(const frieda_proof* p, size_t layer);
const uint32_t* frieda_proof_layer_fri_witness(const frieda_proof* p, size_t layer, size_t* n_qm31);
const uint8_t* frieda_proof_layer_hash_witness(const frieda_proof* p, size_t layer, size_t* n_hashes);
const uint32_t* frieda_proof_layer_column_witness(const frieda_proof* p, size_t layer, size_t* n_m31);
const uint32_t* frieda_proof_last_layer_poly(const frieda_proof* p, size_t* n_qm31);
/* canonical little-endian wire image (layout in DESIGN.md §6).  buf == NULL returns the size needed. */
size_t frieda_proof_serialize(const frieda_proof* p, uint8_t* buf, size_t cap);
int frieda_proof_deserialize(const uint8_t* buf, size_t len, frieda_proof** out);

/* ---- Level B: backend-trait granular operations on device buffers ------------------------------- */
/* Column<T> storage */
int frieda_dev_alloc(frieda_ctx* ctx, size_t bytes, void** d_out);
int frieda_dev_free(frieda_ctx* ctx, void* d);
int frieda_dev_upload(frieda_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int frieda_dev_download(frieda_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* Column::at(index) (stwo core/backend Column<T>; CpuBackend is `Vec<T>` indexing — the accessor frieda's evaluation gather
 * uses, src/proof.rs:62-66): one element of a device column to the host.  Synchronises the ctx stream (it is a debugging /
 * trait-completeness accessor, not a hot path — the prover gathers its openings on the device).
 * frieda_dev_at: M31 column.  frieda_dev_at_secure: QM31 from a SecureColumn d_cols[4][stride] (SoA), out = 4 coordinates. */
int frieda_dev_at(frieda_ctx* ctx, const uint32_t* d_col, size_t index, uint32_t* out);
int frieda_dev_at_secure(frieda_ctx* ctx, const uint32_t* d_cols, size_t stride, size_t index, uint32_t out[4]);
/* ColumnOps::bit_reverse_column (stwo core/backend/mod.rs; the trait surface `CirclePoly::<CpuBackend>::new` and
 * `SecureCirclePoly` are instantiated over at src/utils.rs:21,28 and src/proof.rs:47-52): in place, v[i] <-> v[brev(i)] over
 * log_size bits, for each of ncols columns of 2^log_size words laid out `stride` words apart (ncols = 1: a BaseField
 * column; ncols = 4, stride = 2^log_size: a SecureColumn).  Asynchronous on the ctx stream. */
int frieda_bit_reverse_column(frieda_ctx* ctx, uint32_t* d_cols, size_t stride, uint32_t ncols, uint32_t log_size);

/* codec, src/utils.rs:10-33: d_bytes[len] -> d_coef[n_out] felts (30-bit LSB-first chunks), zero padded
 * to n_out words.  frieda_codec_shape gives F (felts), F' (padded) and L (per-column log size). */
int frieda_codec_shape(size_t len, size_t* n_felts, size_t* n_padded, uint32_t* log_size);
int frieda_unpack30(frieda_ctx* ctx, const void* d_bytes, size_t len, uint32_t* d_coef, size_t n_out);

/* PolyOps::precompute_twiddles(Coset::half_odds(log_domain - 1)): device tables of 2^(log_domain-1) words
 * each (levels N/4, N/8, ..., 1, pad 1) owned by the ctx */
int frieda_precompute_twiddles(frieda_ctx* ctx, uint32_t log_domain, const uint32_t** d_twiddles,
                               const uint32_t** d_inv_twiddles);
/* PolyOps::evaluate on CircleDomain of log size log_domain for ncols polynomials of 2^log_coef coefficients
 * each: d_coef[ncols][2^log_coef] -> d_out[ncols][2^log_domain], bit-reversed order */
int frieda_circle_evaluate(frieda_ctx* ctx, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef,
                           uint32_t log_domain, uint32_t* d_out);

/* ---- reconstruction side (not called by frieda's three functions) ----
 * PolyOps::interpolate generalised to one aligned block of the codeword: d_block[ncols][2^log_coef] holds entries
 * block * 2^log_coef .. (block + 1) * 2^log_coef of each column of the bit-reversed evaluation on the 2^log_domain domain
 * (any block < 2^(log_domain - log_coef), i.e. any 1 / 2^B of the codeword); d_coef[ncols][2^log_coef] receives the
 * coefficients.  block = 0 with log_coef == log_domain is stwo's CpuBackend::interpolate. */
int frieda_circle_interpolate(frieda_ctx* ctx, const uint32_t* d_block, uint32_t ncols, uint32_t log_coef, uint32_t log_domain,
                              uint32_t block, uint32_t* d_coef);
/* inverse of frieda_unpack30 (src/utils.rs:10-19 read backwards): n_felts 30-bit felts -> the first len bytes */
int frieda_pack30(frieda_ctx* ctx, const uint32_t* d_felts, size_t n_felts, void* d_bytes, size_t len);
/* both together for frieda's 4-column layout: a block of the 4 evaluation columns -> the original len bytes */
int frieda_reconstruct_device(frieda_ctx* ctx, const uint32_t* d_block, uint32_t log_coef, uint32_t log_domain, uint32_t block,
                              size_t len, void* d_out_bytes);

/* Reconstruction from scattered cells — what a sampling client holds.  A cell is an aligned run of 2^log_cell consecutive
 * entries (log_cell >= 0; log_cell == 0: single sampled points, README.md:56-69's sample() flow) of the bit-reversed evaluation, the same run of every column: cell c = entries c * 2^log_cell ..
 * (c + 1) * 2^log_cell, c < 2^(log_domain - log_cell).  ANY n_cells = 2^(log_coef - log_cell) distinct cells (at most
 * 2^FRIEDA_MAX_LOG_CELLS = 4096) determine the polynomial: every cell's block transform is undone on the device, then a
 * n_cells x n_cells linear system recombines the coefficient slices (inverted on the host up to 256 cells, by a blocked
 * Gauss-Jordan on the device beyond: cubic in n_cells, ~50 ms at 4096).  d_cells[n_cells][ncols][2^log_cell] (cell-major),
 * cell_index: host array.  n_cells == 1 is frieda_circle_interpolate with block = cell_index[0].  FRIEDA_ERR_ARG for repeated or
 * out-of-range cells (and for a singular system, which distinct cells have never produced). */
int frieda_circle_interpolate_cells(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t ncols,
                                    uint32_t log_cell, uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef);
/* Over-determined form (what a client that sampled a few positions too many holds): n_avail >= 2^(log_coef - log_cell) cells are
 * offered (d_cells[n_avail][ncols][2^log_cell], cell_index[n_avail]; repeated indices are tolerated); the call takes, in the order
 * given, 2^(log_coef - log_cell) of them whose rows of the cell matrix are linearly independent and reconstructs from those — so a
 * point set that happens to be singular (possible with single points, log_cell == 0) costs a spare sample instead of an error.
 * out_used (optional, 2^(log_coef - log_cell) words): positions in the caller's list of the cells used.  At most 256 needed cells
 * (the selection runs on the host); FRIEDA_ERR_ARG when the offered cells do not span the polynomial space.  A client holding at
 * least two points more than the polynomial has coefficients needs neither the selection nor the bound: see
 * frieda_circle_interpolate_points below. */
int frieda_circle_interpolate_cells_any(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_avail, uint32_t ncols,
                                        uint32_t log_cell, uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef, uint32_t* out_used);
/* the same for frieda's 4-column layout, followed by the packer: scattered cells -> the original len bytes */
int frieda_reconstruct_cells_device(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t log_cell,
                                    uint32_t log_coef, uint32_t log_domain, size_t len, void* d_out_bytes);

/* Reconstruction from ANY sufficiently large set of samples, with no bound on their number (the README's sample() flow at blob
 * scale: the reference's 128 KiB `blob` fixture has 2^15 coefficients per column — 2^15 + 2 single sampled points of its 2^19
 * codeword rebuild it).  Same cell layout as above (d_cells[n_cells][ncols][2^log_cell], cell_index[n_cells] on the host; repeated
 * cells are ignored); the distinct points offered must number at least 2^log_coef + 2.  No system is solved: with S the first
 * 2^log_coef + 2 points, V_D the vanishing polynomial of the domain and Z_S the product of the lines through pairs of S, the
 * quotient Z = V_D / Z_S vanishes on everything outside S, Z * p is known on the whole domain, and p follows by transforms and one
 * pointwise division on a disjoint domain (erasure.hip).  With log_cell >= 1, S is the first 2^(log_coef - log_cell) + 1 whole
 * cells and Z_S the product of their cell polynomials pi^(log_cell - 1)(x) - k_cell instead.  Cost: five transforms + ~2^(2 log_coef
 * - log_cell) multiplications below 2^15 coefficients; from there on (single points, or many small cells) Z_S comes from a product
 * tree, O(K log^2 K) (a 15.7 MB blob from 2^20 + 2 points of its 2^24 codeword: 3.2 ms).
 * 1 <= log_coef <= log_domain <= FRIEDA_MAX_LOG_DOMAIN - 1, log_domain >= 2.  Every DISTINCT cell offered,
 * used or not, is then compared with the re-encoded result: samples that are not values of one polynomial of 2^log_coef
 * coefficients give FRIEDA_ERR_ARG (and unspecified d_coef contents) instead of a wrong answer.  A repeated cell index is dropped
 * before that (the first occurrence in the list counts; later ones are neither used nor checked): a caller pooling samples from
 * several peers should de-duplicate by index itself if it wants a conflicting repeat reported.  The first call at a size builds the
 * twiddle tables of the product tree's domains (2^7 .. 2^(log_coef + 1) points, about twice the largest in total, kept by the
 * context): the times quoted are for the calls after it. */
int frieda_circle_interpolate_points(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t ncols,
                                     uint32_t log_cell, uint32_t log_coef, uint32_t log_domain, uint32_t* d_coef);
/* the same for frieda's 4-column layout, followed by the packer: sampled points -> the original len bytes */
int frieda_reconstruct_points_device(frieda_ctx* ctx, const uint32_t* d_cells, const uint32_t* cell_index, uint32_t n_cells, uint32_t log_cell,
                                     uint32_t log_coef, uint32_t log_domain, size_t len, void* d_out_bytes);

/* MerkleOps::commit_on_layer(log_size, prev_layer, columns): d_prev is NULL or 2^(log_size+1) hashes;
 * d_cols is a host array of ncols device column pointers (2^log_size words each); d_out gets 2^log_size
 * 32-byte hashes */
int frieda_merkle_commit_layer(frieda_ctx* ctx, uint32_t log_size, const void* d_prev, const uint32_t* const* d_cols,
                               uint32_t ncols, void* d_out);
/* MerkleProver::commit over 4 equal-length columns d_cols[4][2^log_size]: all layers, leaves first
 * (layer log_size at offset 0, ..., root last; offsets from frieda_merkle_layer_offset), 32*(2^(log_size+1)-1)
 * bytes */
int frieda_merkle_commit(frieda_ctx* ctx, const uint32_t* d_cols, uint32_t log_size, void* d_layers);
size_t frieda_merkle_layer_offset(uint32_t log_size, uint32_t layer_log);
/* root only (no layer is stored): the commit() shape */
int frieda_merkle_root(frieda_ctx* ctx, const uint32_t* d_cols, uint32_t log_size, void* d_root);

/* FriOps::fold_circle_into_line: d_dst[4][N/2] (accumulated: dst*alpha^2 + f') from d_src[4][N] */
int frieda_fold_circle_into_line(frieda_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, uint32_t log_domain,
                                 const uint32_t alpha[4]);
/* FriOps::fold_line: d_src[4][2^line_log] on the line domain reached from half_odds(log_domain-1) by
 * doubling -> d_dst[4][2^(line_log-1)] */
int frieda_fold_line(frieda_ctx* ctx, const uint32_t* d_src, uint32_t line_log, uint32_t log_domain,
                     const uint32_t alpha[4], uint32_t* d_dst);

/* PolyOps::evaluate of the four coordinate columns of a SecureCirclePoly (d_coeffs[4][2^log_size] -> d_evals[4][2^log_domain]) +
 * FriOps::fold_circle_into_line with alpha0 (accumulate_line1 != 0: the trait's form d_line1 = d_line1 * alpha0^2 + fold; 0: d_line1 =
 * fold) + one FriOps::fold_line with alpha1 (d_line2[4][2^(log_domain-2)]), in ONE pass over the evaluation where the shape allows
 * (log_size >= 12, 16-byte aligned buffers), as the three operations otherwise; the results are those of the three separate calls.
 * For callers that hold both folding challenges: a verifier re-executing a round, FRI with fold_step 2, BASELINE configs[1].  The
 * prover of src/proof.rs draws alpha1 only after committing to line 1 (its fused fold + tree launches are inside
 * frieda_commit_and_generate_proof).  The four buffers must not overlap (FRIEDA_ERR_ARG): the one-pass form writes the lines while
 * other workgroups are still writing the evaluation. */
int frieda_circle_evaluate_fold2(frieda_ctx* ctx, const uint32_t* d_coeffs, uint32_t log_size, uint32_t log_domain, uint32_t* d_evals,
                                 const uint32_t alpha0[4], int accumulate_line1, uint32_t* d_line1, const uint32_t alpha1[4],
                                 uint32_t* d_line2);

/* ---- the trait methods frieda's three functions never call (SURVEY.md §8b lists them on the plug-in surface behind `CpuBackend`,
 * src/commit.rs:15-17, src/proof.rs:47-58): an `impl PolyOps / FriOps for HipBackend` needs them ----
 * PolyOps::extend(poly, log_size): d_coef[ncols][2^log_coef] -> d_out[ncols][2^log_size], zero-extended (FRIEDA_ERR_INVARIANT for
 * log_size < log_coef, stwo's assert).  Asynchronous on the ctx stream; the buffers must not overlap. */
int frieda_circle_extend(frieda_ctx* ctx, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, uint32_t log_size, uint32_t* d_out);
/* PolyOps::eval_at_point(poly, point): each of the ncols polynomials d_coef[ncols][2^log_coef] at the circle point (x, y) over
 * the secure field (QM31 coordinates as 4 canonical words each); out[ncols][4] on the host.  Synchronises the ctx stream. */
int frieda_circle_eval_at_point(frieda_ctx* ctx, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, const uint32_t point_x[4],
                                const uint32_t point_y[4], uint32_t* out);
/* FriOps::decompose(eval) -> (g, lambda): d_eval[4][2^log_size] (a SecureColumn in bit-reversed order) split into the part inside
 * the FFT space, d_g[4][2^log_size], and the coefficient lambda of the half-coset vanishing polynomial (out_lambda, host):
 * lambda = (sum of the first half - sum of the second half) / 2^log_size; g = eval -/+ lambda.  Synchronises the ctx stream;
 * d_g may be d_eval. */
int frieda_fri_decompose(frieda_ctx* ctx, const uint32_t* d_eval, uint32_t log_size, uint32_t* d_g, uint32_t out_lambda[4]);

/* GrindOps::grind: smallest nonce with trailing_zeros(mix_u64(digest, nonce)) >= pow_bits */
int frieda_grind(frieda_ctx* ctx, const uint8_t digest[32], uint32_t pow_bits, uint64_t* nonce);

#ifdef __cplusplus
}
#endif
#endif
