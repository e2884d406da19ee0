// kernels.h — host-side launchers of the gfx950 kernels (one translation unit per kernel family).
// Every launcher is asynchronous on the given stream and performs no allocation or synchronisation.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "field.h"

namespace frieda {
namespace k {

// offset of line-twiddle level `lv` in a twiddle table for a circle domain of log size n
inline size_t tw_level_offset(uint32_t n, uint32_t lv) { return ((size_t)1 << (n - 1)) - ((size_t)1 << (n - 1 - lv)); }
__device__ __forceinline__ size_t tw_level_offset_dev(uint32_t n, uint32_t lv) {
    return ((size_t)1 << (n - 1)) - ((size_t)1 << (n - 1 - lv));
}

// Launch context: the stream plus an optional per-kernel timer (HIP events recorded on that same stream around
// every launch; off unless frieda_ctx_set_kernel_timing enabled it).  alg_bytes = the algorithmic (compulsory)
// HBM bytes of the launch by the byte model of SURVEY.md §8d / DESIGN.md §5.
//
// Batches: the commit-phase launchers process `batch` independent blobs of one shape per launch.  Blob b's workspace is the
// single-blob workspace shifted by b * bstride bytes (every device pointer a launcher receives is blob 0's; the kernels add
// blockIdx.y — blockIdx.z in the transforms — times bstride); the DevTranscripts form a contiguous array indexed by b;
// twiddle tables are shared.  batch == 1, bstride == 0 is the single-blob case.
struct KernelTimer;
// Tuning and A/B knobs, PER CONTEXT: a context takes its defaults from the FRIEDA_* environment variables when it is
// created and frieda_ctx_set_option changes one for that context alone; nothing here is process-wide state.  Every knob selects
// another kernel / plan for the SAME result.
struct Tuning {
    uint32_t t5_wide_log = 18;   // FRIEDA_T5_WIDE_LOG: smallest launch (nodes) of the register-subtree kernel
    uint32_t t5_reg3_log = 0;    // FRIEDA_T5_REG3_LOG: leaf / fold launches of >= 2^v nodes stop after their three register levels (0 = never)
    uint32_t t9_max_log = 17;    // FRIEDA_T9_MAX_LOG: largest level-A size of the nine-level kernel
    uint32_t top_max_log = 9;    // FRIEDA_TOP_MAX_LOG: largest hand-over size of the top kernel (9 .. 11)
    uint32_t ntt_cpw = 4;        // FRIEDA_NTT_CPW: columns per workgroup of the generic transform kernel
    uint32_t ntt_cpw_small = 1;  // FRIEDA_NTT_CPW_SMALL: columns per workgroup of fast-kernel launches below 512 tiles
    uint32_t ntt_rep = 2;        // FRIEDA_NTT_REP: the first strided pass as ntt_tile12_rep_kernel: 0 never, 1 wherever the shape allows, 2 from 1024 workgroups on
    bool ntt_no_pad8 = false;    // FRIEDA_NTT_NO_PAD8: no padded / 4-layer fast passes
    bool ntt_no_cp = false;      // FRIEDA_NTT_NO_CP: small fold2 launches as one 256-thread workgroup per tile (not four columns side by side)
    bool ntt_tree_reg_only = false;      // FRIEDA_NTT_TREE_REG_ONLY: the fused encode + leaf launch stops after its five register levels
    bool no_encode_tree_fusion = false;  // FRIEDA_NO_ENCODE_TREE_FUSION: last transform pass and leaf launch as two kernels (commitments too)
    bool encode_tree_fusion_prove = false;  // FRIEDA_ENCODE_TREE_FUSION_PROVE: the fused launch for proofs too (default: commitments only)
    bool no_small_fused = false;         // FRIEDA_NO_SMALL_FUSED: the general path for small domains too
    uint32_t unpack_tiles = 4;           // FRIEDA_UNPACK_TILES: 1, 2, 4 or 8 tiles of 256 quads per unpacker workgroup
    bool intt_generic = false;           // FRIEDA_INTT_GENERIC: every inverse pass through the generic one-column kernel
    uint32_t erasure_tree_min_log = 15;  // FRIEDA_ERASURE_TREE_MIN_LOG: smallest log2(coefficients) whose locator is built by the product tree
    uint32_t tail_run_log = 9;           // FRIEDA_TAIL_RUN_LOG: layers of more than 2^v points use the multi-workgroup kernels
    bool host_decommit = false;          // FRIEDA_HOST_DECOMMIT: openings by the host planner + gather launch
    bool gather_copy = false;            // FRIEDA_GATHER_COPY: fallback path stages lists and results through device memory + copies
    uint32_t test_grind_first_log = 0;   // test hook (frieda_ctx_test_set_grind_first_log): a short first nonce range (0 = off)
    // batch policy (host.h, "batch policy"): workspace bytes a batched call may keep in flight, and the fewest calls a context gets
    uint32_t batch_budget_mb = 0;        // FRIEDA_BATCH_BUDGET_MB: 0 = the default (sixteen proofs of a 2^24 domain, ~43 GB)
    uint32_t tree_skip_log = 18;         // FRIEDA_TREE_SKIP_LOG: a proof's trees of >= 2^v leaves keep their levels from the fourth on only; the
                                         // openings re-hash the two levels above the leaves from the layer's values (decommit.hip)
    uint32_t tree_skip_lone_log = 23;    // FRIEDA_TREE_SKIP_LONE_LOG: the same threshold for a call of ONE blob (the re-hash is a ~15 us chain
                                         // in the decommitment whatever the size; a batch shares it, a lone proof pays it: worth it from 2^23 on)
    uint32_t tp_min_wgs = 768;           // FRIEDA_TP_MIN_WGS: launches of at least this many 256-thread workgroups hash in the throughput form (blake2s.h)
    uint32_t grind_iters = 0;            // FRIEDA_GRIND_ITERS: nonces per lane and claim in the batched grind (window = 256 x this); 0 = by batch size
    uint32_t batch_calls_per_ctx = 1;    // FRIEDA_BATCH_CALLS_PER_CTX: a stream is cut into at least this many calls per context in flight
                                         // (measured, profiles/r05_batch_policy_sweep.txt: 1 beats 2 by 3 % at 2^20 and 30 % at 1 KiB blobs, equal at 2^22 / 2^24)
    // facts about the context's device, recorded at creation (not knobs: tuning_set does not reach them)
    uint64_t device_mem_bytes = 0;       // total device memory (hipMemGetInfo); 0 = unknown (a Tuning without a context): the budget is not clamped
    bool lds_opt_in_ok = true;           // the 68 KB dynamic-LDS opt-in of the transform kernels was granted: gates FRIEDA_NTT_CPW = 4 and the
                                         // four-columns-side-by-side fold2 kernel (ntt.hip), also against a later set_option
    uint64_t test_arena_limit = 0;       // test hook (frieda_ctx_test_set_arena_limit): ensure_arena refuses more than this many bytes (0 = off)
};
// trees of a proof with at least 2^(this) leaves are built without the two levels above their leaves (tree.hip TreeArgs::skip_bc,
// decommit.hip node_from_values, prover.cpp's host-planner fallback): ONE rule for the three places, by the blobs of the call
inline uint32_t tree_skip_threshold(const Tuning& t, uint32_t batch) {
    return batch >= 2 ? t.tree_skip_log : (t.tree_skip_lone_log > t.tree_skip_log ? t.tree_skip_lone_log : t.tree_skip_log);
}
Tuning tuning_from_env();
// `name`: the environment variable's name ("FRIEDA_NTT_REP"); false = unknown name or value out of range
bool tuning_set(Tuning& t, const char* name, long value);
const Tuning& tuning_defaults();  // (for Launch objects built without a context)

struct Launch {
    hipStream_t stream;
    KernelTimer* timer;
    uint32_t batch = 1;
    size_t bstride = 0;
    const Tuning* tune = &tuning_defaults();
};
void timer_begin(KernelTimer* t, hipStream_t s, const char* name, double alg_bytes);
void timer_end(KernelTimer* t, hipStream_t s);
struct Scope {
    const Launch& l;
    Scope(const Launch& l_, const char* name, double alg_bytes) : l(l_) {
        if (l.timer) timer_begin(l.timer, l.stream, name, alg_bytes * l.batch);
    }
    ~Scope() {
        if (l.timer) timer_end(l.timer, l.stream);
    }
};

// scalars describing the domain that the tiny-domain (n < 3) paths need in place of table lookups
struct DomainScalars {
    uint32_t init_x, init_y;          // half_odds(n-1).initial point
    uint32_t inv_init_x, inv_init_y;  // their inverses
};

// ---- codec.hip ----
// src/utils.rs:10-33: bytes -> 30-bit felts, zero padded up to n_out (multiple of 4).  In a batch, blob b's bytes start at
// d_bytes + b * src_bstride (the caller's layout); its felts go to d_out shifted by b * L.bstride bytes.
void unpack30(const Launch& L, const uint8_t* d_bytes, size_t len, uint32_t* d_out, size_t n_out, size_t src_bstride = 0);

// ---- diag.hip ----
// pure-compute Blake2s compression rate of this device (measurement aid; see frieda_ctx_blake2s_ceiling)
// clock (optional, 4 doubles): {leaf clock GHz, leaf SIMD cycles per wave-compression, node clock GHz, node cycles}
int blake2s_ceiling(hipStream_t s, uint32_t* d_scratch, double* leaf_per_s, double* node_per_s, double* clock = nullptr);
size_t blake2s_ceiling_scratch_bytes();

// ---- column.hip ----
// ColumnOps::bit_reverse_column in place on `ncols` columns of 2^log_size words, `stride` words apart (1 = BaseField column,
// 4 = SecureColumn)
void bit_reverse_columns(const Launch& L, uint32_t* d_cols, size_t stride, uint32_t ncols, uint32_t log_size);

// ---- twiddle.hip ----
struct TwiddleSeeds {
    CPoint p0;        // point(initial index of half_odds(n-1))
    CPoint step[32];  // point(step << k)
};
// fills d_tw / d_itw (2^(n-1) words each) for the circle domain of log size n >= 1
// d_scratch8k: 8 KiB of device scratch for the fast path (n >= 12); may be null (slow path)
void gen_twiddles(const Launch& L, uint32_t n, const TwiddleSeeds& seeds, uint32_t* d_tw, uint32_t* d_itw, void* d_scratch8k);

// ---- ntt.hip ----
// d_coef[ncols][coef_stride] (2^L live words per column) -> d_out[ncols][out_stride] (2^n per column)
void circle_evaluate(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n,
                     const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride);
// the same, but only the first 2^out_log entries (L <= out_log <= n) of every column's bit-reversed evaluation are produced
void circle_evaluate_prefix(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n, uint32_t out_log,
                            const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride);

// The same transform with its contiguous last pass fused with leaf hashing: when the shape allows it (4 columns, >= 12 real
// layers, 16-byte aligned buffers) the last launch also produces the tree levels n .. n-6 of the Merkle tree over the 4 columns
// (ENCODE_TREE_LEVELS levels, 2^(n-6) hashes at the top; 5 levels with the FRIEDA_NTT_TREE_REG_ONLY knob) and returns the number of
// levels produced; otherwise it is circle_evaluate and returns 0.
//   sink->layers   non-null: generate_proof shape — the evaluation is written to d_out and every level n-1 .. n-6 is stored at its
//                  leaves-first offset in `layers` (the leaf hashes themselves are never written: nothing reads them);
//   sink->layers   null: commit() shape — the evaluation is NOT written by the last pass (d_out only holds the strided passes'
//                  intermediate), only the hashes of the last level produced go to sink->last_out.
struct EncodeTreeSink {
    uint8_t* layers;
    uint8_t* last_out;
};
constexpr uint32_t ENCODE_TREE_LEVELS = 7;
// the generic transform kernel's 68 KiB of dynamic LDS: opt-in for the current device (once per context, at creation)
hipError_t ntt_opt_in_dynamic_lds();
// PolyOps::evaluate of the four coordinate columns + FriOps::fold_circle_into_line (alpha0; `accumulate`: line1 = line1 * alpha0^2 + fold,
// the trait's form, else line1 = fold) + one FriOps::fold_line (alpha1) — the folds ride in the transform's last pass when the shape
// allows (>= 2^12 coefficients per column, 16-byte aligned buffers: returns true), separate launches otherwise (returns false).
struct EncodeFoldSink {
    const uint32_t* itw;
    uint32_t alpha0[4], alpha1[4];
    bool accumulate;
    uint32_t* line1;
    uint32_t* line2;
};
bool circle_evaluate_fold2(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t L, uint32_t n, const uint32_t* d_tw,
                           DomainScalars ds, uint32_t* d_out, size_t out_stride, const EncodeFoldSink& fs, hipError_t* err);
uint32_t circle_evaluate_into_tree(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n,
                               const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride, const EncodeTreeSink* sink);

// ---- intt.hip (reconstruction side) ----
// block `block` (2^L consecutive bit-reversed evaluations per column) -> the 2^L coefficients per column
void circle_interpolate_block(const Launch& L_, const uint32_t* d_block, size_t in_stride, uint32_t ncols, uint32_t L, uint32_t n,
                              uint32_t block, const uint32_t* d_itw, DomainScalars ds, uint32_t* d_coef, size_t out_stride);
// reconstruction from scattered cells, second half: coef slice u = sum_r vinv[u][r] * w[r] (w[R][ncols][2^m], vinv[R][R])
// vinv_pitch: words between rows of d_vinv (0 = R)
void cells_combine(const Launch& L_, const uint32_t* d_w, const uint32_t* d_vinv, uint32_t R, uint32_t ncols, uint32_t m, uint32_t* d_coef,
                   size_t coef_stride, size_t vinv_pitch = 0);
// the R x R cell matrix inverted on the device (blocked Gauss-Jordan; R a multiple of 32, R <= 4096): see intt.hip
size_t cells_inverse_scratch_bytes(uint32_t R);
uint32_t* cells_inverse_index_buffer(uint8_t* d_scratch, uint32_t R);  // the R-word slot of the scratch block for the cell indices
void cells_matrix_inverse_device(const Launch& L_, const uint32_t* d_cell_index, uint32_t R, uint32_t j_bits, uint32_t m, uint32_t n,
                                 const uint32_t* d_tw, uint8_t* d_scratch, const uint32_t** d_vinv_out, size_t* pitch_out,
                                 const uint32_t** d_state_out);
// inverse of unpack30: felts (30 significant bits each) -> the first `len` bytes of the LSB-first bit stream
void pack30(const Launch& L_, const uint32_t* d_felts, size_t n_felts, uint8_t* d_out, size_t len);

// ---- erasure.hip (reconstruction from any >= 2^L + 2 sampled points: erasure-locator route) ----
// the circle domain of log size n as a generator the kernels evaluate per position: initial point of half_odds(n - 1) and the
// multiples 2^b * step of its step
struct ErasureDomain {
    CPoint init;
    CPoint step_pow[32];
    uint32_t n;
};
// points of the domain at bit-reversed positions d_pos[0 .. count) (d_pos == nullptr: positions 0 .. count - 1)
void erasure_points(const Launch& L_, const ErasureDomain& g, const uint32_t* d_pos, uint32_t count, uint32_t* d_px, uint32_t* d_py);
// lines through the points at positions (d_pos[2a], d_pos[2a + 1]), a < n_lines
void erasure_lines(const Launch& L_, const ErasureDomain& g, const uint32_t* d_pos, uint32_t n_lines, uint32_t* d_la, uint32_t* d_lb,
                   uint32_t* d_lc);
// d_z[t] = product over all lines of line(P_t) — own: with the point's own line (t >> 1) replaced by its tangent derivative at P_t;
// d_zpart: erasure_zpart_chunks(count, n_lines) * count words of scratch
size_t erasure_zpart_chunks(uint32_t count, uint32_t n_lines);
void erasure_zeval(const Launch& L_, const uint32_t* d_px, const uint32_t* d_py, uint32_t count, const uint32_t* d_la, const uint32_t* d_lb,
                   const uint32_t* d_lc, uint32_t n_lines, bool own, uint32_t* d_zpart, uint32_t* d_z);
// d_z[t] = V_D'(P_t) / d_z[t] for the domain of log size n (the locator's values on the known points)
void erasure_known_weights(const Launch& L_, const uint32_t* d_px, const uint32_t* d_py, uint32_t count, uint32_t n, uint32_t* d_z);
// samples in aligned cells of 2^m >= 2 entries: d_kc[c] = pi^(m-1)(x) of cell c's points (its first position d_cell_pos[c]); d_z[t] =
// product over the cells of (pi^(m-1)(x_t) - d_kc[c]) — own: without the point's own cell t >> m; and the known-point weights for that form
void erasure_cellconst(const Launch& L_, const ErasureDomain& g, const uint32_t* d_cell_pos, uint32_t n_cells, uint32_t m, uint32_t* d_kc);
void erasure_zeval_cells(const Launch& L_, const uint32_t* d_px, uint32_t count, uint32_t m, const uint32_t* d_kc, uint32_t n_cells, bool own,
                         uint32_t* d_zpart, uint32_t* d_z);
void erasure_known_weights_cells(const Launch& L_, const uint32_t* d_px, uint32_t count, uint32_t n, uint32_t m, uint32_t* d_z);
// The caller's cell list (d_idx[n_cells], repeats allowed, first occurrence counts; domain_cells = 2^(n - log_cell)) -> the de-duplicated
// lists in list order: d_pos[e] = position of point e in the codeword, d_src[e] = its word offset in the sample buffer
// [n_cells][ncols][2^log_cell]; d_state[0] = number of kept cells, d_state[1] != 0 iff an index was out of range.  Scratch: d_owner
// [domain_cells], d_chunk_sum / d_chunk_off [erasure_sample_lists_chunks(n_cells)], d_first_cell / d_first_row [n_cells]; d_pos / d_src
// [n_cells << log_cell].  erasure_cell_firsts: d_out[i] = d_pos[i << log_cell].
size_t erasure_sample_lists_chunks(uint32_t n_cells);
hipError_t erasure_sample_lists(const Launch& L_, const uint32_t* d_idx, uint32_t n_cells, uint32_t domain_cells, uint32_t ncols, uint32_t log_cell,
                          uint32_t* d_owner, uint32_t* d_chunk_sum, uint32_t* d_chunk_off, uint32_t* d_first_cell, uint32_t* d_first_row,
                          uint32_t* d_state, uint32_t* d_pos, uint32_t* d_src);
void erasure_cell_firsts(const Launch& L_, const uint32_t* d_pos, uint32_t n_cells, uint32_t log_cell, uint32_t* d_out);
// single points, large polynomials: Z_S through a product tree.  Leaves: products of 32 consecutive lines as values on the canonic domain of
// 128 points (d_out[node][128], `nodes` leaves; the lines beyond 32 * nodes — at most a few — go into leaf 0); erasure_pairmul multiplies
// neighbouring nodes' values (d_ext[n_nodes][size] -> d_out[ceil(n_nodes / 2)][size]); erasure_ze: d_ze[t] = V_D(P_t) / d_zs[t] on the
// first `count` <= 2^n points of the next canonic domain; erasure_gather: d_out[t] = d_src[d_pos[t]]
void erasure_lines32(const Launch& L_, const uint32_t* d_px128, const uint32_t* d_py128, const uint32_t* d_la, const uint32_t* d_lb, const uint32_t* d_lc,
                     uint32_t n_lines, uint32_t nodes, uint32_t* d_out);
void erasure_pairmul(const Launch& L_, const uint32_t* d_ext, uint32_t n_nodes, uint32_t size, uint32_t* d_out);
void erasure_ze(const Launch& L_, const ErasureDomain& g1, const uint32_t* d_zs, uint32_t count, uint32_t n, uint32_t* d_ze);
// d_out[c][j] (j < 2^out_log) = the coefficient vector d_in[c][2^(out_log + cnt)] folded along `cnt` <= 4 layers of the domain of log size
// dom_n (twiddle table d_tw): circle_evaluate_prefix of the result gives the same first 2^out_log entries as evaluating d_in itself
void erasure_fold_prefix(const Launch& L_, const uint32_t* d_in, size_t in_stride, uint32_t ncols, uint32_t out_log, uint32_t cnt, const uint32_t* d_tw,
                         uint32_t dom_n, uint32_t* d_out, size_t out_stride);
void erasure_gather(const Launch& L_, const uint32_t* d_src, const uint32_t* d_pos, uint32_t count, uint32_t* d_out);
// d_w[c][d_pos[t]] = d_z[t] * d_cells[d_src[t] + c * 2^log_cell] (d_w zeroed by the caller)
void erasure_scatter(const Launch& L_, const uint32_t* d_cells, const uint32_t* d_src, const uint32_t* d_pos, const uint32_t* d_z, uint32_t count,
                     uint32_t ncols, uint32_t log_cell, uint32_t* d_w, size_t w_stride);
// d_block[c][t] = d_ev[c][t] * d_z[t] / V_D(x = d_px[t]), t < count (V_D of the domain of log size n)
void erasure_divide(const Launch& L_, const uint32_t* d_ev, size_t ev_stride, const uint32_t* d_z, const uint32_t* d_px, uint32_t count, uint32_t ncols,
                    uint32_t n, uint32_t* d_block, size_t block_stride);
// *d_mismatch += samples (per column) that differ from d_ev[c][d_pos[t]]
void erasure_check(const Launch& L_, const uint32_t* d_cells, const uint32_t* d_src, const uint32_t* d_pos, uint32_t count, uint32_t ncols,
                   uint32_t log_cell, const uint32_t* d_ev, size_t ev_stride, uint32_t* d_mismatch);

// ---- merkle.hip ----
// leaves of 4 SoA columns: out[i] = H(c0[i], c1[i], c2[i], c3[i], 0 x 12)
void merkle_leaf4(const Launch& L, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, size_t n,
                  uint8_t* d_out);
// inner nodes without column values: out[i] = H(prev[2i] || prev[2i+1])
void merkle_node(const Launch& L, const uint8_t* d_prev, size_t n, uint8_t* d_out);
// general commit_on_layer (any ncols, optional prev); d_col_ptrs is a device array of ncols column pointers
void merkle_layer_generic(const Launch& L, const uint8_t* d_prev, const uint32_t* const* d_col_ptrs, uint32_t ncols, size_t n,
                          uint8_t* d_out);
// all layers of a tree over 4 columns of 2^m (leaves-first layout as frieda_merkle_layer_offset)
void merkle_tree4(const Launch& L, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, uint32_t m,
                  uint8_t* d_layers);
// root only; d_scratch must hold merkle_root_scratch_bytes(m)
size_t merkle_root_scratch_bytes(uint32_t m);
void merkle_root4(const Launch& L, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, uint32_t m,
                  uint8_t* d_scratch, uint8_t* d_root);
inline size_t merkle_layer_offset(uint32_t log_size, uint32_t layer_log) {
    // leaves first: sum_{l = layer_log+1 .. log_size} 32 * 2^l
    return ((size_t)64 << log_size) - ((size_t)64 << layer_log);
}

// ---- tree.hip (fused commit phase) ----
}  // namespace k
struct DevTranscript;
namespace k {
// tree of the first FRI layer (the 4 evaluation columns): every level above the leaf hashes kept in d_layers (the leaf hashes
// themselves are not written: a FRI decommitment opens both members of every queried pair, so nothing ever reads them; the
// slot in d_layers stays reserved and unwritten); when tr is non-null the finishing kernel mixes the root into the device
// transcript and draws the folding alpha
// tr_init (optional): the initial transcript of every blob in pinned host memory (blob b at tr_init + b * tr_init_pitch bytes);
// the finishing kernel reads it from there and initialises the device transcript itself — no copy in the stream
void tree_first_layer(const Launch& L, const uint32_t* cols, size_t stride, uint32_t m, uint8_t* d_layers, DevTranscript* tr,
                      const DevTranscript* tr_init = nullptr, size_t tr_init_pitch = 0);
// Encode + first tree in one call (commit(): src/commit.rs:16-21; FriProver::commit_first_layer: src/proof.rs:48-52): the
// transform of the 4 coordinate columns with its last pass fused with leaf hashing where the shape allows
// (circle_evaluate_into_tree), then the rest of the tree.  d_layers non-null: every level above the leaf hashes is kept and the
// evaluation is written (generate_proof); null: only the root survives, d_scratch must hold merkle_root_scratch_bytes(n) and
// d_root receives it (commit).  tr / tr_init as tree_first_layer.
// small domains (2^8 .. 2^15 points, <= 2^11 coefficients per column: the reference's 1 KiB - 16 KiB bench inputs): unpack + encode +
// first tree as ONE launch + the top kernel, straight from the blob's bytes (device or page-locked host memory)
bool small_domain_shape(const Tuning& tn, uint32_t Lc, uint32_t n);
// per context at creation: checks the device's LDS against the fused small-domain kernel's need (~94 KB per workgroup) and opts the
// function object in; false = use the general path on this device (Tuning::no_small_fused)
bool small_first_opt_in();
void small_encode_and_first_tree(const Launch& L, const uint8_t* d_data, size_t len, size_t data_stride, uint32_t Lc, uint32_t n,
                                 const uint32_t* d_tw, DomainScalars ds, uint32_t* d_eval, size_t eval_stride, uint8_t* d_layers,
                                 uint8_t* d_scratch, uint8_t* d_root, DevTranscript* tr, const DevTranscript* tr_init, size_t tr_init_pitch);
void encode_and_first_tree(const Launch& L, const uint32_t* d_coef, size_t coef_stride, uint32_t Lc, uint32_t n, const uint32_t* d_tw,
                           DomainScalars ds, uint32_t* d_eval, size_t eval_stride, uint8_t* d_layers, uint8_t* d_scratch, uint8_t* d_root,
                           DevTranscript* tr, const DevTranscript* tr_init = nullptr, size_t tr_init_pitch = 0);
// fold the layer `src` (log size src_log; circle evaluation or line layer) with the alpha in tr into dst_vals and build the
// tree of the folded layer in the same launches (leaf hashes not written, as above); finishes with the channel step
void fold_and_tree(const Launch& L, bool circle, const uint32_t* src, size_t src_stride, uint32_t src_log, uint32_t n,
                   const uint32_t* d_itw, DomainScalars ds, uint32_t* dst_vals, uint8_t* d_layers, DevTranscript* tr);
// all remaining layers (each <= 2048 points) in one workgroup, ending with the last-layer interpolation + mix_felts
constexpr uint32_t TAIL_LOG = 11;
void fri_tail(const Launch& L, const uint32_t* src, size_t src_stride, uint32_t src_log, bool src_is_circle, uint32_t n,
              const uint32_t* d_itw, DomainScalars ds, uint32_t last_log, uint32_t last, uint32_t n_layers, uint32_t* const* vals,
              uint8_t* const* trees, DevTranscript* tr, uint32_t* d_gnext = nullptr);
// proof-of-work scan keyed by tr->ch.digest; atomicMin into tr->nonce
// d_next: L.batch * GRIND_NEXT_STRIDE words of scratch (the per-blob window counters, one 128-byte line each: 2048 workgroups claim
// windows with atomics, and counters sharing a line serialise in one L2 channel; zeroed here unless next_zeroed: fri_tail did it)
constexpr uint32_t GRIND_NEXT_STRIDE = 32;
void grind_dev(const Launch& L, DevTranscript* tr, uint32_t* d_next, uint32_t pow_bits, uint64_t base, uint64_t count,
               bool next_zeroed = false);

// ---- decommit.hip ----
// Query sampling + all openings of a proof in one launch behind the grind (src/proof.rs:59-66).  Per blob the kernel writes,
// into `out + blob * out_stride`:  a header of u32 words [status, n_unique_queries, n_words, n_hashes, |E_1| .. |E_n|], the words
// at words_off (evaluations, then every layer's fri_witness) and the 32-byte hashes at hashes_off (every layer's hash_witness),
// all in proof order.  Layer li has n_witness = |E_{li+1}| and n_hashes = |E_{li+2}| + ... + |E_n|.
constexpr uint32_t DECOMMIT_MAX_QUERIES = 1024;
constexpr uint32_t DECOMMIT_MAX_LOG_DOMAIN = 27;
constexpr uint32_t DECOMMIT_MAX_LAYERS = 40;
constexpr uint32_t DECOMMIT_HEADER_BYTES = 256;
enum DecommitStatus : uint32_t { DECOMMIT_OK = 0, DECOMMIT_NO_NONCE = 1, DECOMMIT_OVERFLOW = 2 };
struct DecommitArgs {
    const DevTranscript* tr;  // array over the blobs: channel state after mix_felts, nonce from the grind
    uint32_t n, n_layers, n_queries;
    size_t bstride;     // bytes between consecutive blobs' workspaces
    uint8_t* out;       // device-visible (normally pinned host) memory
    size_t out_stride;  // bytes between consecutive blobs' output regions
    size_t words_off, hashes_off;
    uint32_t max_words, max_hashes;  // capacity of the two regions (u32 words / hashes) per blob
    const uint32_t* vals[DECOMMIT_MAX_LAYERS];  // blob 0's layers: 4 columns of 2^(n - li) words, column stride 2^(n - li)
    const uint8_t* trees[DECOMMIT_MAX_LAYERS];  // their trees (leaves-first layout)
    uint32_t skip_log;  // trees of >= 2^skip_log leaves: the two levels above the leaves are re-hashed from vals, not read (Tuning::tree_skip_log)
};
void decommit(const Launch& L, const DecommitArgs& a, uint32_t wgs_per_blob);

// ---- polyops.hip: PolyOps::{extend, eval_at_point}, FriOps::decompose (trait completeness; not on frieda's path) ----
struct EvalFactors {
    QM31 f[32];  // f[b]: the factor of coefficient-index bit b (f[0] = point.y, f[1] = point.x, f[b + 1] = 2 f[b]^2 - 1)
};
void circle_extend(const Launch& L, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, uint32_t log_size, uint32_t* d_out);
size_t eval_at_point_scratch_bytes(uint32_t ncols, uint32_t log_coef);
// -> the device address (inside d_scratch) of the ncols x 4 result words, [col][4]
const uint32_t* circle_eval_at_point(const Launch& L, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, const EvalFactors& f, uint32_t* d_scratch);
size_t decompose_scratch_bytes(uint32_t log_size);
// d_eval, d_g: [4][2^log_size]; lambda is left in the first four words of d_scratch
void fri_decompose(const Launch& L, const uint32_t* d_eval, uint32_t log_size, uint32_t* d_g, uint32_t* d_scratch);

// ---- fri.hip ----
struct Alpha {
    uint32_t v[4];
};
// dst[4][N/2] = dst * alpha^2 + (f0 + alpha f1); src[4][N]
void fold_circle_into_line(const Launch& L, uint32_t* d_dst, size_t dst_stride, const uint32_t* d_src, size_t src_stride,
                           uint32_t n, const uint32_t* d_itw, DomainScalars ds, Alpha alpha);
// src[4][2^m] on the line domain of log size m inside a circle domain of log size n -> dst[4][2^(m-1)]
void fold_line(const Launch& L, const uint32_t* d_src, size_t src_stride, uint32_t m, uint32_t n, const uint32_t* d_itw,
               DomainScalars ds, Alpha alpha, uint32_t* d_dst, size_t dst_stride);
// out_words[i] = base_words[word_idx[i]];  out_hashes[i] = 32 bytes at base + 32 * hash_idx[i]
void gather(const Launch& L, const uint32_t* d_base, const uint64_t* d_word_idx, size_t n_words, uint32_t* d_out_words,
            const uint64_t* d_hash_idx, size_t n_hashes, uint8_t* d_out_hashes);
// scans nonces [base, base + count) for trailing_zeros(compress(digest, nonce)) >= pow_bits; atomicMin into *d_result
void grind_scan(const Launch& L, const uint32_t digest[8], uint32_t pow_bits, uint64_t base, uint64_t count,
                unsigned long long* d_result);

}  // namespace k
}  // namespace frieda
