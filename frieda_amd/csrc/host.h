// host.h — host-side data structures of libfrieda_hip: context, proof container, prover / verifier entry points.
// C++ mirror of the Rust types frieda exposes (/root/reference/src/proof.rs:19-26, src/lib.rs:22-44); the C ABI in
// include/frieda_hip.h is a thin shell over these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/frieda_hip.h"
#include "channel.h"
#include "field.h"
#include "kernels.h"

namespace frieda {

using Hash32 = std::array<uint8_t, 32>;

// stwo FriLayerProof { fri_witness, decommitment: MerkleDecommitment { hash_witness, column_witness }, commitment }
struct LayerProof {
    std::vector<QM31> fri_witness;
    std::vector<Hash32> hash_witness;
    std::vector<uint32_t> column_witness;
    Hash32 commitment{};
};

// frieda::proof::Proof (src/proof.rs:19-26) with stwo's FriProof flattened in
struct ProofData {
    LayerProof first_layer;
    std::vector<LayerProof> inner_layers;
    std::vector<QM31> last_layer_poly;  // LinePoly coefficients in stwo's internal (bit-reversed) order
    uint64_t proof_of_work = 0;
    frieda_pcs_config pcs_config{};
    uint32_t log_size_bound = 0;
    std::vector<QM31> evaluations;
};

std::vector<uint8_t> serialize_proof(const ProofData& p);
bool deserialize_proof(const uint8_t* buf, size_t len, ProofData& out);

// ---- circle-group helpers on the host (index arithmetic mod 2^31, stwo core/circle.rs) ----
struct Coset {
    uint32_t initial, step, log_size;
    static Coset half_odds(uint32_t log_size);
    Coset doubled() const;
    uint32_t index_at(uint32_t i) const;
    CPoint at(uint32_t i) const;
};
CPoint point_from_index(uint32_t index);
// coset of the FRI line layer of log size m inside the circle domain of log size n
Coset line_coset(uint32_t n, uint32_t m);

// shape of polynomial_from_bytes (src/utils.rs:21-33) by integer arithmetic
struct CodecShape {
    size_t n_felts, n_padded;
    uint32_t log_size;  // L: per-column log size
};
CodecShape codec_shape(size_t len);

// ---- device context ----
struct TwiddleSet {
    uint32_t* d_tw = nullptr;
    uint32_t* d_itw = nullptr;
    void* d_scratch = nullptr;  // 8 KiB for the generator's point table
    k::DomainScalars ds{};
};

struct ProveJob;  // state of a proof between prove_begin and prove_finish (prover.cpp)
struct ProveJobDeleter {
    void operator()(ProveJob* j) const;
};

// per-kernel HIP-event timer: events are recorded on the ctx stream around each launch; durations are read back
// (after a stream synchronise) by report()
struct KernelTimerImpl;

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool cache_twiddles = true;
    k::Tuning tuning = k::tuning_from_env();  // per-context knobs (kernels.h): environment defaults, frieda_ctx_set_option overrides
    uint32_t test_draw_bound = 2u * P31;  // test hook: acceptance bound of draw_felt (see frieda_ctx_test_set_draw_bound)
    bool host_channel = false;  // evaluate the Fiat-Shamir channel on the host between layers (diagnostic / fallback path)
    std::map<uint32_t, TwiddleSet> twiddles;
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;
    void* pinned = nullptr;  // small pinned staging block for D2H of roots / nonces
    size_t pinned_bytes = 0;
    void* pinned_in = nullptr;  // page-locked input block for lone small host blobs (read by the fused small-domain kernel in place)
    std::string err;
    std::string notes;  // what context creation degraded (refused LDS opt-ins ...), one line each: frieda_ctx_notes — never in `err`
    double phase_ms[8] = {0};  // host wall-clock marks of the last prove() (ms since entry): enqueued, device done, queries, gather, assembled
    KernelTimerImpl* timer = nullptr;  // non-null while kernel timing is enabled
    std::unique_ptr<ProveJob, ProveJobDeleter> job;  // the proof in flight, if any
    uint32_t commit_pending = 0;  // roots of a commit_batch_begin not yet collected by commit_batch_finish (they sit in `pinned`)
    // Fiat-Shamir transcript of the last finished proof (blob 0 of a batch), kept for frieda_ctx_last_transcript
    struct LastTranscript {
        std::vector<Hash32> roots;                      // one per FRI layer (first + inner)
        std::vector<std::array<uint32_t, 4>> alphas;    // the folding challenge drawn after each root
        uint32_t digest_before_grind[8] = {0};          // channel digest after mix_felts(last_layer_poly)
    } last_transcript;

    k::Launch launch() const;
    int set_kernel_timing(bool enabled);
    std::string kernel_timing_report(bool reset);  // JSON; synchronises the stream

    int fail(int code, const std::string& what);
    int hip_fail(hipError_t e, const char* what);
    int ensure_arena(size_t bytes);
    int get_twiddles(uint32_t n, TwiddleSet& out);
    void drop_twiddles();
    ~Ctx();
};

#define FR_HIP(ctx, expr)                                            \
    do {                                                             \
        hipError_t e__ = (expr);                                     \
        if (e__ != hipSuccess) return (ctx)->hip_fail(e__, #expr);   \
    } while (0)

// One proof (or batch) may be in flight per context between prove_begin and prove_finish: its workspace offsets point into the
// arena and its transcript summary sits in the pinned block.  Every other entry point that (re)allocates or writes either of
// them refuses to run meanwhile.
#define FR_NO_JOB(ctx)                                                                                                 \
    do {                                                                                                               \
        if ((ctx)->job || (ctx)->commit_pending)                                                                        \
            return (ctx)->fail(FRIEDA_ERR_ARG, "a proof is in flight on this context (finish it first, or use another context)"); \
    } while (0)

// bump allocator over the ctx arena (sizes are planned before ensure_arena)
struct ArenaPlan {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    }
};

// ---- prover / verifier ----
int commit_device(Ctx* ctx, const uint8_t* d_data, size_t len, uint32_t log_blowup, uint8_t* d_root, bool data_in_arena);
int commit_host(Ctx* ctx, const uint8_t* data, size_t len, uint32_t log_blowup, uint8_t out_root[32]);
int prove(Ctx* ctx, const uint8_t* data, size_t len, bool data_on_device, const uint64_t* seed, frieda_pcs_config cfg,
          uint8_t out_commitment[32], ProofData& out);
// split form: prove_begin enqueues the whole commit phase and returns without synchronising; prove_finish waits for it,
// opens the queries and assembles the proof.  One proof in flight per ctx; use several ctxs to overlap proofs.
int prove_begin(Ctx* ctx, const uint8_t* data, size_t len, bool data_on_device, const uint64_t* seed, frieda_pcs_config cfg);
int prove_finish(Ctx* ctx, uint8_t out_commitment[32], ProofData& out);
// batches: `count` blobs of `len` bytes each (blob b at data + b * data_stride), one seed per blob or null; every kernel of the
// commit phase processes all blobs in one launch (Fiat-Shamir and launch latency are paid once per batch).  Needs the device
// channel (last layer <= 2^11 points).  prove_finish_batch writes count * 32 bytes of commitments.
int prove_begin_batch(Ctx* ctx, const uint8_t* data, size_t data_stride, size_t len, uint32_t count, bool data_on_device,
                      const uint64_t* seeds, frieda_pcs_config cfg, const uint8_t* const* host_ptrs = nullptr);
// `count` separate host blobs of one length
int prove_begin_batch_ptrs(Ctx* ctx, const uint8_t* const* blobs, size_t len, uint32_t count, const uint64_t* seeds, frieda_pcs_config cfg);
int prove_finish_batch(Ctx* ctx, uint8_t* out_commitments, std::vector<ProofData>& outs);
uint32_t job_count(const Ctx* ctx);  // blobs of the job in flight (0: none)
int commit_batch(Ctx* ctx, const uint8_t* data, size_t data_stride, size_t len, uint32_t count, bool data_on_device, uint32_t log_blowup,
                 uint8_t* out_roots, const uint8_t* const* host_ptrs = nullptr);
// split form (frieda_commit_many alternates two contexts: the upload of one unit runs under the kernels of the other): _begin
// enqueues upload, kernels and the download of the roots into the pinned block; _finish waits and copies `count * 32` bytes out
int commit_batch_begin(Ctx* ctx, const uint8_t* data, size_t data_stride, size_t len, uint32_t count, bool data_on_device, uint32_t log_blowup,
                       const uint8_t* const* host_ptrs = nullptr);
int commit_batch_finish(Ctx* ctx, uint8_t* out_roots);
// ---- batch policy: how a run of equal-length blobs is cut into batched calls ("bytes in flight") ----
// Every kernel of a batched call covers all its blobs, so the launch / Fiat-Shamir latency chain is paid once per call: small blobs
// want many per call, large ones few (the workspace grows with the count).  Measured on MI355X (profiles/r04 larger_batch rows, r05):
// what matters is the bytes of workspace a call keeps in flight, not the blob count.
//   per_call = clamp(budget / workspace_bytes_per_blob, 1, ceil(count / (calls_per_ctx * in_flight))),
//   calls    = the smallest multiple of in_flight (no context idles while the last call runs alone) with no call above per_call,
//              sizes equal to within one.
// One implementation for frieda_prove_many / frieda_commit_many (multi.cpp), frieda_batch_plan (callers that drive _begin / _finish
// themselves: frieda_amd.BatchPipeline, bench.py) and the tests.
size_t workspace_bytes_per_blob(size_t len, uint32_t log_blowup, uint32_t log_last_layer, bool prove, bool data_on_device);
// FRIEDA_BATCH_BUDGET_MB, or the default: sixteen proofs of a 2^24 domain (blowup 2^4), ~43 GB — clamped to the device the context sits on:
// the default to 16 % of its memory (two calls in flight: a third; 43 GB are 15 % of an MI355X's 288 GB), an explicit value to 45 %
uint64_t batch_budget_bytes(const k::Tuning& t);
uint32_t batch_per_call(const k::Tuning& t, size_t ws_per_blob, uint32_t count, uint32_t in_flight);
void batch_cut(uint32_t count, uint32_t per_call, uint32_t in_flight, std::vector<uint32_t>& calls);

// returns FRIEDA_OK with *ok set, or FRIEDA_ERR_INVARIANT where the reference panics
// out_queries (optional): on acceptance, the sorted distinct query positions the transcript sampled — evaluations[i] of the proof is
// the value of the 4 columns at position out_queries[i] of the bit-reversed codeword (src/proof.rs:62-66)
int verify(const ProofData& proof, const uint64_t* seed, int* ok, std::vector<uint32_t>* out_queries = nullptr);

// transcript pieces shared by prover and verifier (transcript.cpp)
void channel_mix_felts(Channel& ch, const std::vector<QM31>& felts);
std::vector<uint32_t> generate_queries(Channel& ch, uint32_t log_domain_size, uint32_t n_queries);
std::vector<uint32_t> fold_queries(const std::vector<uint32_t>& q, uint32_t n_folds);

// host Merkle node hash (verifier)
Hash32 hash_node_host(const uint8_t* left, const uint8_t* right, const uint32_t* values, size_t n_values);

}  // namespace frieda

// Recycled proof objects of one context.  A proof of a 2^24 domain is ~140 KB in ~50 vectors; allocating them afresh costs
// 30-40 us per proof (page faults on heap memory the allocator had trimmed after the previous proof was freed) — more than
// copying the openings into them.  frieda_proof_free hands the object back to the pool of the context that made it (the pool
// outlives the context while proofs of it are alive), and the next proof is assembled into the recycled vectors.
struct frieda_proof;
struct ProofPool {
    std::mutex mu;
    std::vector<frieda_proof*> free_list;
    size_t bytes = 0;                                  // approximate capacity held by the free list
    static constexpr size_t MAX_BYTES = 64u << 20;     // beyond this, freed proofs are really freed
    static constexpr size_t MAX_ENTRIES = 4096;
    frieda_proof* get();            // a recycled object (contents unspecified, capacities kept) or a new one; never null (throws bad_alloc)
    void put(frieda_proof* p);      // takes ownership
    ~ProofPool();
};
struct frieda_ctx {
    frieda::Ctx c;
    std::shared_ptr<ProofPool> pool = std::make_shared<ProofPool>();
};

// the extern "C" entry points' exception fence: no exception crosses the ABI
#define FR_GUARD_BEGIN try {
#define FR_GUARD_END(ctxptr)                                          \
    }                                                                 \
    catch (const std::bad_alloc&) {                                   \
        if (ctxptr) (ctxptr)->c.err = "host allocation failed";       \
        return FRIEDA_ERR_NOMEM;                                      \
    }                                                                 \
    catch (const std::exception& e) {                                 \
        if (ctxptr) (ctxptr)->c.err = e.what();                       \
        return FRIEDA_ERR_INVARIANT;                                  \
    }
struct frieda_proof {
    frieda::ProofData p;
    std::shared_ptr<ProofPool> home;  // set while the object is out with the caller; null inside the pool and for clones / parsed proofs
};
