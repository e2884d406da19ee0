// prover.cpp — commit() and commit_and_generate_proof() on the device.
//
// Orchestrates the gfx950 kernels in the order of the reference call stack:
//   commit   (/root/reference/src/commit.rs:11-22): codec -> twiddles -> 4 x circle FFT -> Merkle root
//   prove    (/root/reference/src/proof.rs:32-77; stwo core/fri.rs FriProver::{commit, decommit}):
//            codec -> FFT -> first tree -> [mix root, draw alpha, fold, tree]* -> last layer interpolate ->
//            grind -> queries -> decommit (one gather launch) -> Proof
// The blob enters as raw bytes (3.75 B per felt over PCIe) and is unpacked on the device; after that nothing but the
// 2 KB transcript summary (roots, last-layer polynomial, nonce, channel state) and the gathered openings ever cross PCIe.
// The Fiat–Shamir channel runs inside the commit-phase kernels (dev_transcript.h); the host synchronises twice per proof:
// after the grind and after the gather.  prove_begin / prove_finish expose that split so several proofs can overlap.
// (Host-side channel between layers is kept as a policy: frieda_ctx_set_host_channel, and for last layers above 2^11 points.)
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include <stddef.h>

#include "dev_transcript.h"
#include "host.h"

namespace frieda {

namespace {

void hash_to_words(const uint8_t* h, uint32_t (&w)[8]) {
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)h[4 * i] | ((uint32_t)h[4 * i + 1] << 8) | ((uint32_t)h[4 * i + 2] << 16) | ((uint32_t)h[4 * i + 3] << 24);
}

int ensure_pinned(Ctx* ctx, size_t bytes) {
    if (ctx->pinned_bytes >= bytes) return FRIEDA_OK;
    if (ctx->pinned) FR_HIP(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    FR_HIP(ctx, hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_bytes = bytes;
    return FRIEDA_OK;
}

// Small host blobs (the fused small-domain kernel, tree.hip) are not copied to the device at all: they are placed in a page-locked
// block that the kernel reads directly (one PCIe read per workgroup instead of a copy command in front of the first launch).
constexpr size_t SMALL_HOST_IN_BYTES = 7680;  // 15 << 9: up to 32 workgroups read the block over PCIe; larger small blobs are copied to the device first
int ensure_pinned_in(Ctx* ctx) {
    if (ctx->pinned_in) return FRIEDA_OK;
    FR_HIP(ctx, hipHostMalloc(&ctx->pinned_in, SMALL_HOST_IN_BYTES, hipHostMallocDefault));
    return FRIEDA_OK;
}

// validated domain shape shared by commit and prove
struct Shape {
    CodecShape cs;
    uint32_t L, B, n;
    size_t N;
};

int make_shape(Ctx* ctx, size_t len, uint32_t B, Shape& sh) {
    // B arrives unchecked from the caller (a ctypes c_uint32 turns -1 into 0xFFFFFFFF): bound it before any sum is formed
    if (B > FRIEDA_MAX_LOG_DOMAIN) return ctx->fail(FRIEDA_ERR_ARG, "log_blowup_factor larger than FRIEDA_MAX_LOG_DOMAIN");
    if (len > ((size_t)1 << 40)) return ctx->fail(FRIEDA_ERR_ARG, "blob larger than 2^40 bytes");
    sh.cs = codec_shape(len);
    sh.L = sh.cs.log_size;
    sh.B = B;
    // Coset::half_odds(L + B - 1) panics for L + B == 0 (u32 underflow)
    if ((uint64_t)sh.L + B < 1) return ctx->fail(FRIEDA_ERR_INVARIANT, "log_size + log_blowup_factor must be >= 1");
    if ((uint64_t)sh.L + B > FRIEDA_MAX_LOG_DOMAIN) return ctx->fail(FRIEDA_ERR_ARG, "domain larger than FRIEDA_MAX_LOG_DOMAIN");
    sh.n = sh.L + B;
    sh.N = (size_t)1 << sh.n;
    return FRIEDA_OK;
}

// LineEvaluation::interpolate for the last FRI layer (stwo core/poly/line.rs), host side: <= 2^14 points.
// `v`: evaluations in bit-reversed order on LineDomain(c); returns coefficients in LinePoly's internal order.
void line_interpolate(std::vector<QM31>& v, Coset c) {
    const uint32_t lg = c.log_size;
    const size_t n = v.size();
    for (uint32_t i = 0; i < n; i++) {
        uint32_t j = bit_reverse(i, lg);
        if (i < j) std::swap(v[i], v[j]);
    }
    for (Coset d = c; d.log_size > 0; d = d.doubled()) {
        const size_t sz = (size_t)1 << d.log_size, half = sz / 2;
        std::vector<uint32_t> ix(half);
        CPoint p = point_from_index(d.initial), st = point_from_index(d.step);
        for (size_t i = 0; i < half; i++) {
            ix[i] = m31_inv(p.x);
            p = cp_add(p, st);
        }
        for (size_t base = 0; base < n; base += sz)
            for (size_t i = 0; i < half; i++) {
                QM31 a = v[base + i], b = v[base + half + i];
                v[base + i] = qm_add(a, b);
                v[base + half + i] = qm_scale(qm_sub(a, b), ix[i]);
            }
    }
    uint32_t len_inv = m31_inv((uint32_t)n);
    for (auto& q : v) q = qm_scale(q, len_inv);
}

void bit_reverse_vec(std::vector<QM31>& v, size_t count, uint32_t lg) {
    for (uint32_t i = 0; i < count; i++) {
        uint32_t j = bit_reverse(i, lg);
        if (i < j) std::swap(v[i], v[j]);
    }
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// commit
// -------------------------------------------------------------------------------------------------
int commit_device(Ctx* ctx, const uint8_t* d_data, size_t len, uint32_t log_blowup, uint8_t* d_root, bool data_in_arena) {
    FR_NO_JOB(ctx);
    Shape sh;
    int rc = make_shape(ctx, len, log_blowup, sh);
    if (rc) return rc;
    FR_HIP(ctx, hipSetDevice(ctx->device));

    ArenaPlan plan;
    size_t o_data = plan.take(data_in_arena ? len : 0);
    size_t o_coef = plan.take(sizeof(uint32_t) * sh.cs.n_padded);
    size_t o_eval = plan.take(sizeof(uint32_t) * 4 * sh.N);
    size_t o_scr = plan.take(k::merkle_root_scratch_bytes(sh.n));
    (void)o_data;
    if (!data_in_arena) {
        rc = ctx->ensure_arena(plan.off);
        if (rc) return rc;
    } else if (plan.off > ctx->arena_bytes) {
        return ctx->fail(FRIEDA_ERR_ARG, "internal: arena not sized by caller");
    }
    TwiddleSet tw;
    rc = ctx->get_twiddles(sh.n, tw);
    if (rc) return rc;

    uint32_t* coef = reinterpret_cast<uint32_t*>(ctx->arena + o_coef);
    uint32_t* eval = reinterpret_cast<uint32_t*>(ctx->arena + o_eval);
    if (k::small_domain_shape(ctx->tuning, sh.L, sh.n)) {
        k::small_encode_and_first_tree(ctx->launch(), d_data, len, 0, sh.L, sh.n, tw.d_tw, tw.ds, nullptr, 0, nullptr, ctx->arena + o_scr, d_root,
                                       nullptr, nullptr, 0);
    } else {
        k::unpack30(ctx->launch(), d_data, len, coef, sh.cs.n_padded);
        k::encode_and_first_tree(ctx->launch(), coef, (size_t)1 << sh.L, sh.L, sh.n, tw.d_tw, tw.ds, eval, sh.N, nullptr, ctx->arena + o_scr, d_root,
                                 nullptr);
    }
    FR_HIP(ctx, hipGetLastError());
    return FRIEDA_OK;
}

int commit_host(Ctx* ctx, const uint8_t* data, size_t len, uint32_t log_blowup, uint8_t out_root[32]) {
    FR_NO_JOB(ctx);
    Shape sh;
    int rc = make_shape(ctx, len, log_blowup, sh);
    if (rc) return rc;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    // same plan as commit_device(data_in_arena = true), plus 32 bytes for the root
    ArenaPlan plan;
    size_t o_data = plan.take(len);
    plan.take(sizeof(uint32_t) * sh.cs.n_padded);
    plan.take(sizeof(uint32_t) * 4 * sh.N);
    plan.take(k::merkle_root_scratch_bytes(sh.n));
    size_t o_root = plan.take(32);
    rc = ctx->ensure_arena(plan.off);
    if (rc) return rc;
    rc = ensure_pinned(ctx, 4096);
    if (rc) return rc;
    if (k::small_domain_shape(ctx->tuning, sh.L, sh.n) && len <= SMALL_HOST_IN_BYTES) {
        // no copy commands at all: the kernels read the blob from page-locked host memory and write the root into it
        rc = ensure_pinned_in(ctx);
        if (rc) return rc;
        if (len) memcpy(ctx->pinned_in, data, len);
        rc = commit_device(ctx, static_cast<const uint8_t*>(ctx->pinned_in), len, log_blowup, static_cast<uint8_t*>(ctx->pinned), true);
        if (rc) return rc;
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(out_root, ctx->pinned, 32);
        return FRIEDA_OK;
    }
    if (len) FR_HIP(ctx, hipMemcpyAsync(ctx->arena + o_data, data, len, hipMemcpyHostToDevice, ctx->stream));
    rc = commit_device(ctx, ctx->arena + o_data, len, log_blowup, ctx->arena + o_root, true);
    if (rc) return rc;
    FR_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->arena + o_root, 32, hipMemcpyDeviceToHost, ctx->stream));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out_root, ctx->pinned, 32);
    return FRIEDA_OK;
}

// commit() of `count` blobs of one length in one pass of launches (blob b at data + b * data_stride); roots to host memory
int commit_batch(Ctx* ctx, const uint8_t* data, size_t data_stride, size_t len, uint32_t count, bool data_on_device, uint32_t log_blowup,
                 uint8_t* out_roots, const uint8_t* const* host_ptrs) {
    int rc = commit_batch_begin(ctx, data, data_stride, len, count, data_on_device, log_blowup, host_ptrs);
    if (rc) return rc;
    return commit_batch_finish(ctx, out_roots);
}

int commit_batch_finish(Ctx* ctx, uint8_t* out_roots) {
    if (!ctx->commit_pending) return ctx->fail(FRIEDA_ERR_ARG, "no commit batch in flight on this context");
    const uint32_t count = ctx->commit_pending;
    ctx->commit_pending = 0;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FR_HIP(ctx, hipGetLastError());
    memcpy(out_roots, ctx->pinned, 32 * (size_t)count);
    return FRIEDA_OK;
}

int commit_batch_begin(Ctx* ctx, const uint8_t* data, size_t data_stride, size_t len, uint32_t count, bool data_on_device, uint32_t log_blowup,
                       const uint8_t* const* host_ptrs) {
    FR_NO_JOB(ctx);
    if (count == 0 || count > 65535) return ctx->fail(FRIEDA_ERR_ARG, "batch count out of range");
    if (count > 1 && data_stride < len) return ctx->fail(FRIEDA_ERR_ARG, "batch stride smaller than the blob length");
    Shape sh;
    int rc = make_shape(ctx, len, log_blowup, sh);
    if (rc) return rc;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    ArenaPlan plan;
    const size_t o_data = plan.take(data_on_device ? 0 : len);
    const size_t o_coef = plan.take(sizeof(uint32_t) * sh.cs.n_padded);
    const size_t o_eval = plan.take(sizeof(uint32_t) * 4 * sh.N);
    const size_t o_scr = plan.take(k::merkle_root_scratch_bytes(sh.n));
    const size_t o_root = plan.take(32);
    const size_t bstride = plan.off;
    rc = ctx->ensure_arena(bstride * count);
    if (rc) return rc;
    rc = ensure_pinned(ctx, 32 * (size_t)count + 4096);
    if (rc) return rc;
    TwiddleSet tw;
    rc = ctx->get_twiddles(sh.n, tw);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    uint8_t* A = ctx->arena;
    k::Launch LN = ctx->launch();
    LN.batch = count;
    LN.bstride = count > 1 ? bstride : 0;
    const uint8_t* d_data = data;
    size_t d_stride = data_stride;
    if (!data_on_device) {
        if (len && host_ptrs) {  // separate host blobs of one length (frieda_commit_many)
            for (uint32_t b = 0; b < count; b++) FR_HIP(ctx, hipMemcpyAsync(A + o_data + (size_t)b * bstride, host_ptrs[b], len, hipMemcpyHostToDevice, s));
        } else if (len) {
            FR_HIP(ctx, hipMemcpy2DAsync(A + o_data, bstride, data, count > 1 ? data_stride : len, len, count, hipMemcpyHostToDevice, s));
        }
        d_data = A + o_data;
        d_stride = bstride;
    }
    uint32_t* coef = reinterpret_cast<uint32_t*>(A + o_coef);
    uint32_t* eval = reinterpret_cast<uint32_t*>(A + o_eval);
    if (k::small_domain_shape(ctx->tuning, sh.L, sh.n)) {
        k::small_encode_and_first_tree(LN, d_data, len, d_stride, sh.L, sh.n, tw.d_tw, tw.ds, nullptr, 0, nullptr, A + o_scr, A + o_root, nullptr,
                                       nullptr, 0);
    } else {
        k::unpack30(LN, d_data, len, coef, sh.cs.n_padded, d_stride);
        k::encode_and_first_tree(LN, coef, (size_t)1 << sh.L, sh.L, sh.n, tw.d_tw, tw.ds, eval, sh.N, nullptr, A + o_scr, A + o_root, nullptr);
    }
    FR_HIP(ctx, hipMemcpy2DAsync(ctx->pinned, 32, A + o_root, bstride, 32, count, hipMemcpyDeviceToHost, s));
    FR_HIP(ctx, hipGetLastError());
    ctx->commit_pending = count;
    return FRIEDA_OK;
}

// -------------------------------------------------------------------------------------------------
// commit_and_generate_proof
// -------------------------------------------------------------------------------------------------
namespace {

struct FriLayerDev {
    size_t o_vals;   // 4 columns of 2^log words
    size_t o_tree;   // leaves-first layers
    uint32_t log;
};

// Collects the device reads of one layer's decommitment in proof order
struct GatherPlan {
    std::vector<uint64_t> word_idx;  // u32 index from the arena base
    std::vector<uint64_t> hash_idx;  // 32-byte index from the arena base
};

// compute_decommitment_positions_and_witness_evals (fold_step = 1): positions + requests for the witness values
std::vector<uint32_t> plan_witness(const std::vector<uint32_t>& queries, const FriLayerDev& lay, GatherPlan& g, size_t& n_witness) {
    std::vector<uint32_t> pos;
    n_witness = 0;
    const size_t stride = (size_t)1 << lay.log;
    for (size_t i = 0; i < queries.size();) {
        size_t j = i;
        while (j < queries.size() && (queries[j] >> 1) == (queries[i] >> 1)) j++;
        uint32_t start = (queries[i] >> 1) << 1;
        size_t kq = i;
        for (uint32_t p = start; p < start + 2; p++) {
            pos.push_back(p);
            if (kq < j && queries[kq] == p) {
                kq++;
                continue;
            }
            for (int c = 0; c < 4; c++) g.word_idx.push_back(lay.o_vals / 4 + (size_t)c * stride + p);
            n_witness++;
        }
        i = j;
    }
    return pos;
}

// MerkleProver::decommit for a tree with its columns on the leaf layer only: requests for the sibling hashes.
// stwo walks the layers from the leaves up; on layer l it visits, in increasing order, every node that has a queried leaf or a
// visited child below it, and emits the hash of each child that was not visited itself (left before right).  `positions` are
// the queried leaves (sorted, unique).  Two scratch vectors are reused across calls (this runs ~25 times per proof).
size_t plan_merkle_decommit(const std::vector<uint32_t>& positions, const FriLayerDev& lay, GatherPlan& g) {
    static thread_local std::vector<uint32_t> buf_a, buf_b;
    std::vector<uint32_t>* below = &buf_a;  // visited nodes of the layer below (children), sorted
    std::vector<uint32_t>* here = &buf_b;
    below->assign(positions.begin(), positions.end());  // leaf layer: exactly the queried leaves, no hashes needed
    size_t n_hashes = 0;
    for (int layer = (int)lay.log - 1; layer >= 0; layer--) {
        const size_t child_base = (lay.o_tree + k::merkle_layer_offset(lay.log, (uint32_t)layer + 1)) / 32;
        here->clear();
        const uint32_t* c = below->data();
        const size_t nc = below->size();
        for (size_t i = 0; i < nc;) {
            const uint32_t node = c[i] >> 1;
            const bool has_left = c[i] == 2 * node;
            if (has_left) i++;
            const bool has_right = i < nc && c[i] == 2 * node + 1;
            if (has_right) i++;
            if (!has_left) {
                g.hash_idx.push_back(child_base + 2 * (size_t)node);
                n_hashes++;
            }
            if (!has_right) {
                g.hash_idx.push_back(child_base + 2 * (size_t)node + 1);
                n_hashes++;
            }
            here->push_back(node);
        }
        std::swap(below, here);
    }
    return n_hashes;
}

}  // namespace

// State of a proof between prove_begin (commit phase enqueued) and prove_finish (decommit + assembly).
struct ProveJob {
    frieda_pcs_config cfg{};
    Shape sh{};
    uint32_t n = 0, last = 0, last_log = 0, n_inner = 0;
    size_t N = 0;
    FriLayerDev first{};
    std::vector<FriLayerDev> inner;
    size_t o_lastv = 0, o_nonce = 0, o_tr = 0, o_gnext = 0, o_widx = 0, o_hidx = 0, o_wout = 0, o_hout = 0;
    size_t max_words = 0, max_hashes = 0, tr_host_pitch = 0;
    bool dev_channel = false;
    // openings on the device (decommit.hip): per-blob output regions in the pinned block behind the transcript summaries
    bool dev_decommit = false;
    size_t dec_off = 0, dec_stride = 0, dec_words_off = 0, dec_hashes_off = 0;
    uint32_t dec_max_words = 0, dec_max_hashes = 0;
    // a batch of `count` blobs of one shape: blob b's workspace is blob 0's shifted by b * bstride bytes; the transcripts
    // (o_tr) and the gather regions sit behind the workspaces
    uint32_t count = 1;
    size_t bstride = 0;
    // the tree-skip threshold the trees of THIS job were built with (k::tree_skip_threshold at begin): the openings — device kernel
    // and host-planner fallback — must use the same one even if frieda_ctx_set_option moved the knob between _begin and _finish
    uint32_t skip_log = 0;
    struct Blob {
        Channel ch{};
        std::vector<Hash32> roots;
        std::vector<std::array<uint32_t, 4>> alphas;  // per layer, for frieda_ctx_last_transcript
        std::vector<QM31> lastv;
        uint64_t nonce = ~0ull;
        uint32_t digest_before_grind[8] = {0};
    };
    std::vector<Blob> blobs;
    uint64_t grind_base = 0, grind_chunk = 0;
    std::chrono::steady_clock::time_point t_start;
};

void ProveJobDeleter::operator()(ProveJob* j) const { delete j; }

namespace {
// the per-blob part of a proof's workspace (one plan for the prover and for the batch policy, which divides its budget by plan.off)
void plan_prove_blob(ArenaPlan& plan, const Shape& sh, uint32_t last_log, size_t host_len, size_t& o_data, size_t& o_coef, FriLayerDev& first,
                     std::vector<FriLayerDev>& inner, size_t& o_lastv, size_t& o_nonce) {
    const uint32_t n = sh.n, n_inner = (n - 1) - last_log;
    o_data = plan.take(host_len);
    o_coef = plan.take(sizeof(uint32_t) * sh.cs.n_padded);
    first = FriLayerDev{plan.take(sizeof(uint32_t) * 4 * sh.N), 0, n};
    first.o_tree = plan.take(k::merkle_layer_offset(n, 0) + 32);
    inner.resize(n_inner);
    for (uint32_t kx = 0; kx < n_inner; kx++) {
        uint32_t lg = n - 1 - kx;
        inner[kx].log = lg;
        inner[kx].o_vals = plan.take(sizeof(uint32_t) * 4 << lg);
        inner[kx].o_tree = plan.take(k::merkle_layer_offset(lg, 0) + 32);
    }
    o_lastv = plan.take(sizeof(uint32_t) * 4 << last_log);
    o_nonce = plan.take(8);
}
}  // namespace

// Device workspace one blob of `len` bytes adds to a batched call (what ensure_arena is asked for, per blob): the figure the batch
// policy (batch_plan below) divides its budget by.  0: the shape is outside what the prover accepts.
size_t workspace_bytes_per_blob(size_t len, uint32_t log_blowup, uint32_t log_last_layer, bool prove, bool data_on_device) {
    if (log_blowup > FRIEDA_MAX_LOG_DOMAIN || len > ((size_t)1 << 40)) return 0;
    Shape sh;
    sh.cs = codec_shape(len);
    sh.L = sh.cs.log_size;
    sh.B = log_blowup;
    if ((uint64_t)sh.L + log_blowup < 1 || (uint64_t)sh.L + log_blowup > FRIEDA_MAX_LOG_DOMAIN) return 0;
    sh.n = sh.L + log_blowup;
    sh.N = (size_t)1 << sh.n;
    ArenaPlan plan;
    if (!prove) {  // commit_batch_begin's plan
        plan.take(data_on_device ? 0 : len);
        plan.take(sizeof(uint32_t) * sh.cs.n_padded);
        plan.take(sizeof(uint32_t) * 4 * sh.N);
        plan.take(k::merkle_root_scratch_bytes(sh.n));
        plan.take(32);
        return plan.off;
    }
    if (log_last_layer > 10 || sh.n < 2 || sh.L < 1 + log_last_layer || log_last_layer + log_blowup > sh.n - 1) return 0;
    size_t a, b, c, d;
    FriLayerDev first;
    std::vector<FriLayerDev> inner;
    plan_prove_blob(plan, sh, log_last_layer + log_blowup, data_on_device ? 0 : len, a, b, first, inner, c, d);
    return plan.off;
}

// ---- batch policy (host.h) ----
uint64_t batch_budget_bytes(const k::Tuning& t) {
    //  the figure below was measured on a 288 GB device; on a smaller one it must not ask for what the device does not have.
    // What is FREE right now is the callers' business (multi.cpp asks hipMemGetInfo before it cuts a device's run, and halves on NOMEM).
    if (t.batch_budget_mb) {
        const uint64_t b = (uint64_t)t.batch_budget_mb << 20;
        return t.device_mem_bytes ? std::min<uint64_t>(b, t.device_mem_bytes / 100 * 45) : b;
    }
    if (t.device_mem_bytes) return std::min<uint64_t>(batch_budget_bytes(k::tuning_defaults()), t.device_mem_bytes / 100 * 16);  // (16 %: a device that reports 288e9 bytes still gets the sixteen proofs the figure was measured with)
    // measured at the headline size and stated in workspace bytes so that it carries over to every other size: sixteen proofs of a
    // 2^24 domain per call (15 MiB blobs: 2^22 felts -> 2^20 coefficients per column, blowup 2^4), ~43 GB per call in flight — two
    // calls in flight hold 30 % of a 288 GB MI355X.  With the compression's throughput form a call's fixed part (the narrow launches
    // of its latency chain, ~1.6 ms at this size) weighs more than it used to: 1.89 ms per blob at 5 per call, 1.85 at 15, 1.83 at 30
    // (profiles/r05_batch_policy_sweep.txt, second block)
    static const uint64_t sixteen_headline = 16 * (uint64_t)workspace_bytes_per_blob((size_t)15 << 20, 4, 0, true, true);
    return sixteen_headline;
}

uint32_t batch_per_call(const k::Tuning& t, size_t ws_per_blob, uint32_t count, uint32_t in_flight) {
    if (count == 0) return 1;
    if (in_flight == 0) in_flight = 1;
    uint64_t per = ws_per_blob ? batch_budget_bytes(t) / ws_per_blob : 1;
    const uint64_t ways = (uint64_t)std::max<uint32_t>(1, t.batch_calls_per_ctx) * in_flight;
    const uint64_t spread = (count + ways - 1) / ways;  // every context gets its calls
    per = std::min<uint64_t>(per, spread);
    // (the batched entry points take up to 65535 blobs; beyond a few thousand per call the latency chain is amortised to nothing, while
    // the pinned staging block — ~10 KB of openings per proof — and the host-side proof objects keep growing with the count)
    per = std::min<uint64_t>(per, 4096);
    return (uint32_t)std::max<uint64_t>(per, 1);
}

void batch_cut(uint32_t count, uint32_t per_call, uint32_t in_flight, std::vector<uint32_t>& calls) {
    calls.clear();
    if (count == 0) return;
    if (per_call == 0) per_call = 1;
    if (in_flight == 0) in_flight = 1;
    // the number of calls is a multiple of the calls in flight: an odd call out would run alone on the chip, its latency chain and
    // its launches' ramps un-overlapped (20 blobs: 4 x 5, not 5 x 4); no call exceeds per_call (the budget is a ceiling)
    const uint64_t round = (uint64_t)per_call * in_flight;
    uint64_t n = (uint64_t)in_flight * ((count + round - 1) / round);
    n = std::min<uint64_t>(n, count);
    const uint32_t base = (uint32_t)(count / n), extra = (uint32_t)(count % n);
    calls.assign((size_t)n, base);
    for (uint32_t i = 0; i < extra; i++) calls[i] = base + 1;
}


static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// transcript summaries (header + last-layer polynomial) of all blobs -> pinned memory, blob b at b * tr_host_pitch
static int download_transcripts(Ctx* ctx, const ProveJob& J) {
    const size_t bytes = offsetof(DevTranscript, last_poly) + ((size_t)16 << J.last);
    const uint8_t* d_tr = ctx->arena + J.o_tr;
    if (J.count == 1) {
        FR_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_tr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        FR_HIP(ctx, hipMemcpy2DAsync(ctx->pinned, J.tr_host_pitch, d_tr, sizeof(DevTranscript), bytes, J.count, hipMemcpyDeviceToHost, ctx->stream));
    }
    return FRIEDA_OK;
}

// decommit.hip over all blobs of the job: reads the device transcripts (nonce, channel) and the layers, writes the pinned block
static void launch_decommit(Ctx* ctx, const ProveJob& J, const k::Launch& LN) {
    uint8_t* A = ctx->arena;
    k::DecommitArgs a{};
    a.tr = reinterpret_cast<const DevTranscript*>(A + J.o_tr);
    a.n = J.n;
    a.n_layers = 1 + J.n_inner;
    a.n_queries = J.cfg.n_queries;
    a.bstride = J.count > 1 ? J.bstride : 0;
    a.out = static_cast<uint8_t*>(ctx->pinned) + J.dec_off;
    a.out_stride = J.dec_stride;
    a.words_off = J.dec_words_off;
    a.hashes_off = J.dec_hashes_off;
    a.max_words = J.dec_max_words;
    a.max_hashes = J.dec_max_hashes;
    a.skip_log = J.skip_log;
    a.vals[0] = reinterpret_cast<const uint32_t*>(A + J.first.o_vals);
    a.trees[0] = A + J.first.o_tree;
    for (uint32_t kx = 0; kx < J.n_inner; kx++) {
        a.vals[1 + kx] = reinterpret_cast<const uint32_t*>(A + J.inner[kx].o_vals);
        a.trees[1 + kx] = A + J.inner[kx].o_tree;
    }
    // a lone large proof has ~4000 hashes to fetch: several workgroups share them (each repeats the cheap table set-up)
    const uint32_t wgs = J.count >= 8 ? 1u : (J.count >= 2 ? 2u : (J.n >= 16 ? 8u : 2u));
    k::decommit(LN, a, wgs);
}

int prove_begin(Ctx* ctx, const uint8_t* data, size_t len, bool data_on_device, const uint64_t* seed, frieda_pcs_config cfg) {
    return prove_begin_batch(ctx, data, 0, len, 1, data_on_device, seed, cfg);
}

// the same for `count` separate HOST blobs of one length (blob b at blobs[b]): what frieda_prove_many hands a device
int prove_begin_batch_ptrs(Ctx* ctx, const uint8_t* const* blobs, size_t len, uint32_t count, const uint64_t* seeds, frieda_pcs_config cfg) {
    if (!blobs || count == 0) return ctx->fail(FRIEDA_ERR_ARG, "null blob table");
    return prove_begin_batch(ctx, blobs[0], len, len, count, false, seeds, cfg, blobs);
}

// `count` blobs of `len` bytes each, blob b at data + b * data_stride (or, host blobs only, at host_ptrs[b]); seeds: null or one per blob
int prove_begin_batch(Ctx* ctx, const uint8_t* data, size_t data_stride, size_t len, uint32_t count, bool data_on_device,
                      const uint64_t* seeds, frieda_pcs_config cfg, const uint8_t* const* host_ptrs) {
    const auto t_entry = std::chrono::steady_clock::now();
    FR_NO_JOB(ctx);  // a proof or a commit batch in flight owns the arena and the pinned block
    if (count == 0 || count > 65535) return ctx->fail(FRIEDA_ERR_ARG, "batch count out of range");
    if (count > 1 && data_stride < len) return ctx->fail(FRIEDA_ERR_ARG, "batch stride smaller than the blob length");
    const uint32_t B = cfg.log_blowup_factor, last = cfg.log_last_layer_degree_bound;
    // both arrive unchecked: bound them before `last + B` and the shifts that use it are formed
    if (last > 10) return ctx->fail(FRIEDA_ERR_ARG, "log_last_layer_degree_bound > 10");
    if (B > FRIEDA_MAX_LOG_DOMAIN) return ctx->fail(FRIEDA_ERR_ARG, "log_blowup_factor larger than FRIEDA_MAX_LOG_DOMAIN");
    std::unique_ptr<ProveJob, ProveJobDeleter> jp(new ProveJob());
    ProveJob& J = *jp;
    Shape& sh = J.sh;
    int rc = make_shape(ctx, len, B, sh);
    if (rc) return rc;
    // FriProver::commit_last_layer asserts evaluation.len() == last_layer_domain_size: needs L - 1 >= last;
    // the line domain half_odds(n - 1) needs n >= 2
    if (sh.n < 2 || sh.L < 1 + last) return ctx->fail(FRIEDA_ERR_INVARIANT, "polynomial too small for the FRI configuration");
    if (cfg.n_queries == 0 || cfg.n_queries > 4096) return ctx->fail(FRIEDA_ERR_ARG, "n_queries out of range");
    if (cfg.pow_bits > 48) return ctx->fail(FRIEDA_ERR_ARG, "pow_bits > 48");
    FR_HIP(ctx, hipSetDevice(ctx->device));
    J.cfg = cfg;
    J.last = last;
    J.count = count;
    J.skip_log = k::tree_skip_threshold(ctx->tuning, count);
    J.blobs.resize(count);
    const uint32_t n = J.n = sh.n, last_log = J.last_log = last + B;
    const size_t N = J.N = sh.N;
    const uint32_t n_inner = J.n_inner = (n - 1) - last_log;

    // ---- workspace plan ----
    ArenaPlan plan;
    size_t o_data = 0, o_coef = 0;
    FriLayerDev& first = J.first;
    std::vector<FriLayerDev>& inner = J.inner;
    plan_prove_blob(plan, sh, last_log, data_on_device ? 0 : len, o_data, o_coef, first, inner, J.o_lastv, J.o_nonce);
    const size_t o_lastv = J.o_lastv, o_nonce = J.o_nonce;
    // everything above is per blob; what follows is shared by the batch
    const size_t bstride = J.bstride = plan.off;  // a multiple of 256
    plan.off = bstride * count;
    const size_t o_tr = J.o_tr = plan.take(sizeof(DevTranscript) * count);
    J.o_gnext = plan.take(sizeof(uint32_t) * k::GRIND_NEXT_STRIDE * count);  // grind window counters, one cache line per blob
    // decommit gather: per layer <= 2 positions per query; hashes <= 2 * queries * log per layer
    J.max_words = (size_t)cfg.n_queries * 4 * (1 + (n_inner + 1)) * count;
    J.max_hashes = (size_t)cfg.n_queries * 2 * (size_t)(n + 1) * (n_inner + 1) * count;
    J.o_widx = plan.take(8 * J.max_words);
    J.o_hidx = plan.take(8 * J.max_hashes);
    J.o_wout = plan.take(4 * J.max_words);
    J.o_hout = plan.take(32 * J.max_hashes);
    rc = ctx->ensure_arena(plan.off);
    if (rc) return rc;
    // pinned staging: the transcript summaries of all blobs (header + last-layer polynomial each), or the last layer itself on
    // the host-channel path, or the gather lists and results
    const size_t tr_host_pitch = J.tr_host_pitch = (offsetof(DevTranscript, last_poly) + ((size_t)16 << last) + 63) & ~(size_t)63;
    // device-side openings: header + words + hashes per blob, behind the transcript summaries
    J.dec_max_words = (uint32_t)(J.max_words / count);
    J.dec_max_hashes = (uint32_t)(J.max_hashes / count);
    J.dec_words_off = k::DECOMMIT_HEADER_BYTES;
    J.dec_hashes_off = J.dec_words_off + ((4 * (size_t)J.dec_max_words + 15) & ~(size_t)15);
    J.dec_stride = (J.dec_hashes_off + 32 * (size_t)J.dec_max_hashes + 255) & ~(size_t)255;
    J.dec_off = (tr_host_pitch * count + 255) & ~(size_t)255;
    const size_t pinned_need =
        std::max<size_t>(std::max<size_t>(std::max<size_t>(tr_host_pitch * count, (sizeof(uint32_t) * 4) << last_log),
                                          (4 + 8) * J.max_words + (32 + 8) * J.max_hashes + 512),
                         J.dec_off + J.dec_stride * count);
    rc = ensure_pinned(ctx, pinned_need);
    if (rc) return rc;
    TwiddleSet tw;
    rc = ctx->get_twiddles(n, tw);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    J.t_start = std::chrono::steady_clock::now();
    for (double& v : ctx->phase_ms) v = 0.0;
    k::Launch LN = ctx->launch();  // every launch below covers the whole batch
    LN.batch = count;
    LN.bstride = count > 1 ? bstride : 0;
    uint8_t* A = ctx->arena;
    // The whole commit phase runs on the device (transcript included) when the last layer fits the single-workgroup tail;
    // otherwise (log_last_layer_degree_bound + log_blowup_factor > 11) the channel is evaluated on the host between layers.
    const bool dev_channel = J.dev_channel = last_log <= k::TAIL_LOG && !ctx->host_channel;
    if (count > 1 && !dev_channel)
        return ctx->fail(FRIEDA_ERR_ARG, "batches need the device channel (last layer <= 2^11 points, host channel policy off)");

    // ---- encode (src/proof.rs:38,44-50) ----
    const uint8_t* d_data = data;
    size_t d_data_stride = data_stride;
    const bool small = k::small_domain_shape(ctx->tuning, sh.L, n) && !ctx->host_channel && last_log <= k::TAIL_LOG;
    if (!data_on_device) {
        if (small && count == 1 && len <= SMALL_HOST_IN_BYTES) {
            // a lone small blob is not copied to the device: the first kernel reads it from page-locked host memory
            rc = ensure_pinned_in(ctx);
            if (rc) return rc;
            const uint8_t* src = host_ptrs ? host_ptrs[0] : data;
            if (len) memcpy(ctx->pinned_in, src, len);
            d_data = static_cast<const uint8_t*>(ctx->pinned_in);
            d_data_stride = 0;
        } else {
            if (len && host_ptrs) {
                for (uint32_t b = 0; b < count; b++) FR_HIP(ctx, hipMemcpyAsync(A + o_data + (size_t)b * bstride, host_ptrs[b], len, hipMemcpyHostToDevice, s));
            } else {
                if (len && count == 1) FR_HIP(ctx, hipMemcpyAsync(A + o_data, data, len, hipMemcpyHostToDevice, s));
                if (len && count > 1) FR_HIP(ctx, hipMemcpy2DAsync(A + o_data, bstride, data, data_stride, len, count, hipMemcpyHostToDevice, s));
            }
            d_data = A + o_data;
            d_data_stride = bstride;
        }
    }
    uint32_t* coef = reinterpret_cast<uint32_t*>(A + o_coef);
    uint32_t* eval = reinterpret_cast<uint32_t*>(A + first.o_vals);
    if (!small) k::unpack30(LN, d_data, len, coef, sh.cs.n_padded, d_data_stride);
    ctx->phase_ms[5] = ms_since(t_entry);  // set-up before the first launch (workspace plan, twiddle lookup, ...) + that launch call

    for (uint32_t b = 0; b < count; b++) {
        J.blobs[b].ch.init();
        if (seeds) J.blobs[b].ch.mix_u64(seeds[b]);  // src/proof.rs:40-42
        J.blobs[b].roots.assign(1 + n_inner, Hash32{});
    }
    Channel& ch = J.blobs[0].ch;  // (the host-channel path below is single-blob)
    const uint64_t* seed = seeds;
    (void)seed;

    auto cols = [&](const FriLayerDev& lay, int c) { return reinterpret_cast<uint32_t*>(A + lay.o_vals) + ((size_t)c << lay.log); };
    std::vector<Hash32>& roots = J.blobs[0].roots;
    std::vector<QM31>& lastv = J.blobs[0].lastv;
    if (dev_channel) {
        DevTranscript* d_tr = reinterpret_cast<DevTranscript*>(A + o_tr);
        const size_t hdr = offsetof(DevTranscript, last_poly);
        {
            // initial transcripts, staged back to back at tr_host_pitch and copied into the device array in one go
            for (uint32_t b = 0; b < count; b++) {
                DevTranscript* ht = reinterpret_cast<DevTranscript*>(static_cast<uint8_t*>(ctx->pinned) + b * tr_host_pitch);
                memset(ht, 0, hdr);
                ht->ch = J.blobs[b].ch;
                ht->nonce = ~0ull;
                ht->draw_bound = ctx->test_draw_bound;
            }
            // no copy in the stream: the kernel that finishes the first tree reads these few words from the pinned block itself
            // (the block is next written by the transcript download behind the grind, i.e. after that kernel)
        }
        // the encode's last pass runs fused with FriProver::commit_first_layer's leaf hashing (src/proof.rs:48-52); small domains: unpack +
        // encode + first tree in one launch, straight from the blob
        if (small)
            k::small_encode_and_first_tree(LN, d_data, len, d_data_stride, sh.L, n, tw.d_tw, tw.ds, eval, N, A + first.o_tree, nullptr, nullptr, d_tr,
                                           reinterpret_cast<const DevTranscript*>(ctx->pinned), tr_host_pitch);
        else
            k::encode_and_first_tree(LN, coef, (size_t)1 << sh.L, sh.L, n, tw.d_tw, tw.ds, eval, N, A + first.o_tree, nullptr, nullptr, d_tr,
                                     reinterpret_cast<const DevTranscript*>(ctx->pinned), tr_host_pitch);
        // FriProver::commit_inner_layers: layers above 2^11 points, one fused fold + tree each
        const FriLayerDev* cur = &first;
        bool circle = true;
        uint32_t kx = 0;
        // layers of more than 2^TAIL_RUN_LOG points go through the multi-workgroup kernels (a 2^11 layer is faster there than in
        // the one-workgroup tail); FRIEDA_TAIL_RUN_LOG is a tuning knob
        const uint32_t tail_run_log = ctx->tuning.tail_run_log;
        while (kx < n_inner && inner[kx].log > tail_run_log) {
            k::fold_and_tree(LN, circle, cols(*cur, 0), (size_t)1 << cur->log, cur->log, n, tw.d_itw, tw.ds, cols(inner[kx], 0),
                             A + inner[kx].o_tree, d_tr);
            cur = &inner[kx];
            circle = false;
            kx++;
        }
        // everything that is left, the last layer's interpolation and mix_felts included, in one workgroup
        uint32_t* tvals[16];
        uint8_t* ttrees[16];
        const uint32_t n_tail = (n_inner - kx) + 1;
        if (n_tail > 16) return ctx->fail(FRIEDA_ERR_INVARIANT, "internal: tail too long");
        for (uint32_t i = 0; i + 1 < n_tail; i++) {
            tvals[i] = cols(inner[kx + i], 0);
            ttrees[i] = A + inner[kx + i].o_tree;
        }
        tvals[n_tail - 1] = reinterpret_cast<uint32_t*>(A + o_lastv);
        ttrees[n_tail - 1] = nullptr;
        uint32_t* d_gnext = reinterpret_cast<uint32_t*>(A + J.o_gnext);
        k::fri_tail(LN, cols(*cur, 0), (size_t)1 << cur->log, cur->log, circle, n, tw.d_itw, tw.ds, last_log, last, n_tail, tvals, ttrees,
                    d_tr, d_gnext);
        // grind (src/proof.rs:58), keyed by the digest now sitting in the device transcript: first chunk + transcript download
        J.grind_base = 0;
        // first range: 16x the expected search (a miss has probability e^-16; workgroups without work leave at once)
        J.grind_chunk = (uint64_t)1 << std::max<uint32_t>(22, std::min<uint32_t>(cfg.pow_bits, 36) + 4);
        if (ctx->tuning.test_grind_first_log) J.grind_chunk = (uint64_t)1 << ctx->tuning.test_grind_first_log;  // test hook: the retry loop runs
        k::grind_dev(LN, d_tr, d_gnext, cfg.pow_bits, J.grind_base, J.grind_chunk, /*next_zeroed=*/true);
        rc = download_transcripts(ctx, J);
        if (rc) return rc;
        // mix_u64(nonce), query sampling and every opening of the proof, written in proof order into the pinned block
        J.dev_decommit = cfg.n_queries <= k::DECOMMIT_MAX_QUERIES && 1 + n_inner <= k::DECOMMIT_MAX_LAYERS && n <= k::DECOMMIT_MAX_LOG_DOMAIN &&
                         !ctx->tuning.host_decommit;
        if (J.dev_decommit) launch_decommit(ctx, J, LN);
        FR_HIP(ctx, hipGetLastError());
        ctx->phase_ms[0] = ms_since(J.t_start);  // commit phase fully enqueued
    } else {
        auto fetch_root = [&](const FriLayerDev& lay, Hash32& root) -> int {
            FR_HIP(ctx, hipMemcpyAsync(ctx->pinned, A + lay.o_tree + k::merkle_layer_offset(lay.log, 0), 32, hipMemcpyDeviceToHost, s));
            FR_HIP(ctx, hipStreamSynchronize(s));
            memcpy(root.data(), ctx->pinned, 32);
            return FRIEDA_OK;
        };

        // ---- FriProver::commit_first_layer ----
        k::circle_evaluate(LN, coef, (size_t)1 << sh.L, 4, sh.L, n, tw.d_tw, tw.ds, eval, N);
        k::merkle_tree4(LN, cols(first, 0), cols(first, 1), cols(first, 2), cols(first, 3), n, A + first.o_tree);
        rc = fetch_root(first, roots[0]);
        if (rc) return rc;
        uint32_t rw[8];
        hash_to_words(roots[0].data(), rw);
        ch.mix_root(rw);

        // ---- FriProver::commit_inner_layers ----
        QM31 alpha = ch.draw_felt(ctx->test_draw_bound);
        J.blobs[0].alphas.assign(1 + n_inner, std::array<uint32_t, 4>{});
        J.blobs[0].alphas[0] = {alpha.a, alpha.b, alpha.c, alpha.d};
        {
            // LineEvaluation::new_zero then fold_circle_into_line
            uint32_t* dst = (n_inner > 0) ? cols(inner[0], 0) : reinterpret_cast<uint32_t*>(A + o_lastv);
            FR_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(uint32_t) * 4 << (n - 1), s));
            k::fold_circle_into_line(LN, dst, (size_t)1 << (n - 1), eval, N, n, tw.d_itw, tw.ds, k::Alpha{{alpha.a, alpha.b, alpha.c, alpha.d}});
        }
        for (uint32_t kx = 0; kx < n_inner; kx++) {
            const FriLayerDev& lay = inner[kx];
            k::merkle_tree4(LN, cols(lay, 0), cols(lay, 1), cols(lay, 2), cols(lay, 3), lay.log, A + lay.o_tree);
            rc = fetch_root(lay, roots[kx + 1]);
            if (rc) return rc;
            hash_to_words(roots[kx + 1].data(), rw);
            ch.mix_root(rw);
            alpha = ch.draw_felt(ctx->test_draw_bound);
            J.blobs[0].alphas[kx + 1] = {alpha.a, alpha.b, alpha.c, alpha.d};
            uint32_t* dst = (kx + 1 < n_inner) ? cols(inner[kx + 1], 0) : reinterpret_cast<uint32_t*>(A + o_lastv);
            k::fold_line(LN, cols(lay, 0), (size_t)1 << lay.log, lay.log, n, tw.d_itw, tw.ds, k::Alpha{{alpha.a, alpha.b, alpha.c, alpha.d}}, dst,
                         (size_t)1 << (lay.log - 1));
        }

        // ---- FriProver::commit_last_layer ----
        const size_t n_lastdom = (size_t)1 << last_log;
        FR_HIP(ctx, hipMemcpyAsync(ctx->pinned, A + o_lastv, sizeof(uint32_t) * 4 * n_lastdom, hipMemcpyDeviceToHost, s));
        FR_HIP(ctx, hipStreamSynchronize(s));
        lastv.assign(n_lastdom, QM31{0, 0, 0, 0});
        {
            const uint32_t* h = reinterpret_cast<const uint32_t*>(ctx->pinned);
            for (size_t i = 0; i < n_lastdom; i++) lastv[i] = {h[i], h[n_lastdom + i], h[2 * n_lastdom + i], h[3 * n_lastdom + i]};
        }
        line_interpolate(lastv, line_coset(n, last_log));
        bit_reverse_vec(lastv, n_lastdom, last_log);  // into_ordered_coefficients
        const size_t n_poly = (size_t)1 << last;
        for (size_t i = n_poly; i < n_lastdom; i++)
            if (!qm_is_zero(lastv[i])) return ctx->fail(FRIEDA_ERR_INVARIANT, "invalid degree");  // assert! upstream
        lastv.resize(n_poly);
        bit_reverse_vec(lastv, n_poly, last);  // LinePoly::from_ordered_coefficients
        channel_mix_felts(ch, lastv);
        memcpy(J.blobs[0].digest_before_grind, ch.digest, 32);

        // ---- grind (src/proof.rs:58-59) ----
        unsigned long long* d_nonce = reinterpret_cast<unsigned long long*>(A + o_nonce);
        {
            FR_HIP(ctx, hipMemsetAsync(d_nonce, 0xFF, 8, s));
            uint64_t base = 0, chunk = (uint64_t)1 << 22;
            for (;;) {
                k::grind_scan(LN, ch.digest, cfg.pow_bits, base, chunk, d_nonce);
                FR_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_nonce, 8, hipMemcpyDeviceToHost, s));
                FR_HIP(ctx, hipStreamSynchronize(s));
                memcpy(&J.blobs[0].nonce, ctx->pinned, 8);
                if (J.blobs[0].nonce != ~0ull) break;
                base += chunk;
                if (chunk < ((uint64_t)1 << 28)) chunk <<= 1;
            }
        }
        ctx->phase_ms[0] = ms_since(J.t_start);
    }
    ctx->job = std::move(jp);
    return FRIEDA_OK;
}

int prove_finish(Ctx* ctx, uint8_t out_commitment[32], ProofData& out) {
    if (ctx->job && ctx->job->count != 1) return ctx->fail(FRIEDA_ERR_ARG, "the proof in flight is a batch: use the batch finish");
    std::vector<ProofData> outs(1);
    outs[0] = std::move(out);  // assemble into the caller's object: a recycled one keeps its vectors' capacity
    int rc = prove_finish_batch(ctx, out_commitment, outs);
    if (!outs.empty()) out = std::move(outs[0]);
    return rc;
}

uint32_t job_count(const Ctx* ctx) { return ctx->job ? ctx->job->count : 0; }

int prove_finish_batch(Ctx* ctx, uint8_t* out_commitments, std::vector<ProofData>& outs) {
    if (!ctx->job) return ctx->fail(FRIEDA_ERR_ARG, "no proof in flight on this context");
    std::unique_ptr<ProveJob, ProveJobDeleter> jp = std::move(ctx->job);  // released on every exit path
    ProveJob& J = *jp;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    k::Launch LN = ctx->launch();
    LN.batch = J.count;
    LN.bstride = J.count > 1 ? J.bstride : 0;
    uint8_t* A = ctx->arena;
    const frieda_pcs_config cfg = J.cfg;
    const uint32_t n = J.n, last = J.last, n_inner = J.n_inner, count = J.count;
    const size_t N = J.N;
    const FriLayerDev& first = J.first;
    const std::vector<FriLayerDev>& inner = J.inner;

    if (J.dev_channel) {
        DevTranscript* d_tr = reinterpret_cast<DevTranscript*>(A + J.o_tr);
        const size_t n_poly = (size_t)1 << last;
        auto host_tr = [&](uint32_t b) { return reinterpret_cast<const DevTranscript*>(static_cast<const uint8_t*>(ctx->pinned) + b * J.tr_host_pitch); };
        for (;;) {
            FR_HIP(ctx, hipStreamSynchronize(s));
            bool all_found = true;
            for (uint32_t b = 0; b < count; b++) {
                if (host_tr(b)->status & 1u) return ctx->fail(FRIEDA_ERR_INVARIANT, "invalid degree");  // assert! upstream
                all_found &= host_tr(b)->nonce != ~0ull;
            }
            if (all_found) break;
            // next range for the blobs still searching (the others leave at once: their minimum is below every new nonce)
            J.grind_base += J.grind_chunk;
            if (J.grind_chunk < ((uint64_t)1 << 28)) J.grind_chunk <<= 1;
            k::grind_dev(LN, d_tr, reinterpret_cast<uint32_t*>(A + J.o_gnext), cfg.pow_bits, J.grind_base, J.grind_chunk);
            int rc = download_transcripts(ctx, J);
            if (rc) return rc;
            if (J.dev_decommit) launch_decommit(ctx, J, LN);
        }
        FR_HIP(ctx, hipGetLastError());
        for (uint32_t b = 0; b < count; b++) {
            const DevTranscript* ht = host_tr(b);
            ProveJob::Blob& bl = J.blobs[b];
            if (ht->n_roots != 1 + n_inner || ht->n_last_poly != n_poly) return ctx->fail(FRIEDA_ERR_INVARIANT, "internal: transcript out of step");
            bl.nonce = ht->nonce;
            bl.ch = ht->ch;
            memcpy(bl.digest_before_grind, ht->ch.digest, 32);  // downloaded behind the grind, ahead of the decommit kernel
            bl.alphas.resize(1 + n_inner);
            for (uint32_t li = 0; li <= n_inner; li++) {
                for (int w = 0; w < 8; w++)
                    for (int bb = 0; bb < 4; bb++) bl.roots[li][4 * w + bb] = (uint8_t)(ht->roots[li][w] >> (8 * bb));
                bl.alphas[li] = {ht->alphas[li][0], ht->alphas[li][1], ht->alphas[li][2], ht->alphas[li][3]};
            }
            bl.lastv.resize(n_poly);
            for (size_t i = 0; i < n_poly; i++)
                bl.lastv[i] = {ht->last_poly[4 * i], ht->last_poly[4 * i + 1], ht->last_poly[4 * i + 2], ht->last_poly[4 * i + 3]};
        }
    }
    ctx->phase_ms[1] = ms_since(J.t_start);  // commit phase + grind complete on the device (first synchronise)
    ctx->last_transcript.roots = J.blobs[0].roots;
    ctx->last_transcript.alphas = J.blobs[0].alphas;
    memcpy(ctx->last_transcript.digest_before_grind, J.blobs[0].digest_before_grind, 32);

    struct LayerCounts {
        size_t n_witness, n_hashes;
    };
    std::vector<LayerCounts> counts((size_t)count * (1 + n_inner));
    std::vector<size_t> n_evals(count);
    std::vector<const uint32_t*> words_of(count);  // per blob: evaluations, then every layer's witness, in proof order
    std::vector<const uint8_t*> hashes_of(count);  // per blob: every layer's hash witness, in proof order

    // ---- FriProver::decommit ----
    // Normally the device has already done it (decommit.hip, launched behind the grind): the pinned block holds, per blob, a
    // header with the list sizes and the openings in proof order.  Otherwise (more than 1024 queries, an opening table too
    // large for the kernel, FRIEDA_HOST_DECOMMIT) the host draws the queries, plans the openings and runs one gather launch.
    bool from_device = J.dev_decommit;
    if (from_device) {
        for (uint32_t b = 0; b < count && from_device; b++) {
            const uint8_t* reg = static_cast<const uint8_t*>(ctx->pinned) + J.dec_off + (size_t)b * J.dec_stride;
            const uint32_t* hdr = reinterpret_cast<const uint32_t*>(reg);
            if (hdr[0] == k::DECOMMIT_OVERFLOW) {
                from_device = false;
                break;
            }
            if (hdr[0] != k::DECOMMIT_OK) return ctx->fail(FRIEDA_ERR_INVARIANT, "internal: device decommit out of step with the grind");
            n_evals[b] = hdr[1];
            words_of[b] = reinterpret_cast<const uint32_t*>(reg + J.dec_words_off);
            hashes_of[b] = reg + J.dec_hashes_off;
            LayerCounts* cnt = &counts[(size_t)b * (1 + n_inner)];
            // layer li: n_witness = |E_{li+1}|, n_hashes = |E_{li+2}| + ... + |E_n|   (hdr[3 + s] = |E_s|)
            size_t suffix[40] = {0};  // suffix[s] = |E_s| + ... + |E_n|, s <= n + 1 <= 33
            for (uint32_t sft = n; sft >= 1; sft--) suffix[sft] = suffix[sft + 1] + hdr[3 + sft];
            for (uint32_t li = 0; li <= n_inner; li++) {
                cnt[li].n_witness = hdr[3 + li + 1];
                cnt[li].n_hashes = li + 2 <= n ? suffix[li + 2] : 0;
            }
        }
        ctx->phase_ms[2] = ctx->phase_ms[3] = ms_since(J.t_start);
    }
    if (!from_device) {
        // The host plans against COMPLETE trees; the large ones were built without the two levels above their leaves (the device
        // decommitment re-hashes those, tree.hip TreeArgs::skip_bc): rebuild such a tree from its layer's values first (a rare path).
        {
            k::Launch Lb = ctx->launch();  // single-blob launches
            for (uint32_t b = 0; b < count; b++) {
                const size_t boff = (size_t)b * J.bstride;
                auto rebuild = [&](const FriLayerDev& lay) {
                    if (lay.log < J.skip_log) return;
                    const uint32_t* c0 = reinterpret_cast<const uint32_t*>(A + lay.o_vals + boff);
                    const size_t cs = (size_t)1 << lay.log;
                    k::merkle_tree4(Lb, c0, c0 + cs, c0 + 2 * cs, c0 + 3 * cs, lay.log, A + lay.o_tree + boff);
                };
                rebuild(first);
                for (uint32_t kx = 0; kx < n_inner; kx++) rebuild(inner[kx]);
            }
        }
        static thread_local GatherPlan g;  // capacity is kept from proof to proof
        g.word_idx.clear();
        g.hash_idx.clear();
        std::vector<size_t> w_begin(count), h_begin(count);
        for (uint32_t b = 0; b < count; b++) {
            ProveJob::Blob& bl = J.blobs[b];
            bl.ch.mix_u64(bl.nonce);  // src/proof.rs:59
            const std::vector<uint32_t> queries = generate_queries(bl.ch, n, cfg.n_queries);
            n_evals[b] = queries.size();
            w_begin[b] = g.word_idx.size();
            h_begin[b] = g.hash_idx.size();
            const size_t boff = (size_t)b * J.bstride;  // this blob's workspace (a multiple of 256 bytes)
            auto shifted = [&](FriLayerDev lay) {
                lay.o_vals += boff;
                lay.o_tree += boff;
                return lay;
            };
            LayerCounts* cnt = &counts[(size_t)b * (1 + n_inner)];
            // Proof.evaluations (src/proof.rs:62-66)
            for (uint32_t q : queries)
                for (int c = 0; c < 4; c++) g.word_idx.push_back((first.o_vals + boff) / 4 + (size_t)c * N + q);
            {
                const FriLayerDev lay = shifted(first);
                std::vector<uint32_t> pos = plan_witness(queries, lay, g, cnt[0].n_witness);
                cnt[0].n_hashes = plan_merkle_decommit(pos, lay, g);
            }
            std::vector<uint32_t> lq = fold_queries(queries, 1);
            for (uint32_t kx = 0; kx < n_inner; kx++) {
                const FriLayerDev lay = shifted(inner[kx]);
                std::vector<uint32_t> pos = plan_witness(lq, lay, g, cnt[kx + 1].n_witness);
                cnt[kx + 1].n_hashes = plan_merkle_decommit(pos, lay, g);
                lq = fold_queries(lq, 1);
            }
        }
        ctx->phase_ms[2] = ms_since(J.t_start);  // queries drawn, openings planned
        if (g.word_idx.size() > J.max_words || g.hash_idx.size() > J.max_hashes) return ctx->fail(FRIEDA_ERR_INVARIANT, "internal: gather plan overflow");
        // one upload (word indices then hash indices, staged in pinned memory), one launch, one download
        const size_t nw = g.word_idx.size(), nh = g.hash_idx.size();
        const size_t wbytes = 4 * nw, out_bytes = wbytes + 32 * nh;
        uint8_t* hp = reinterpret_cast<uint8_t*>(ctx->pinned);
        uint64_t* hidx = reinterpret_cast<uint64_t*>(hp + ((4 * J.max_words + 32 * J.max_hashes + 255) & ~(size_t)255));
        memcpy(hidx, g.word_idx.data(), 8 * nw);
        memcpy(hidx + nw, g.hash_idx.data(), 8 * nh);
        // the index and output regions of the arena are laid out back to back (words region, then hashes region), so the
        // hash part may start right behind the words actually used
        k::Launch L1 = ctx->launch();  // the gather works on absolute indices: a single-blob launch
        if (out_bytes <= ((size_t)1 << 20) && !ctx->tuning.gather_copy) {
            // small openings (the usual case): the kernel reads its index lists from, and writes its results to, the pinned
            // staging block directly over PCIe — one launch instead of copy + launch + copy (each copy costs 10-20 us of setup)
            k::gather(L1, reinterpret_cast<const uint32_t*>(A), hidx, nw, reinterpret_cast<uint32_t*>(hp), hidx + nw, nh, hp + wbytes);
        } else {
            FR_HIP(ctx, hipMemcpyAsync(A + J.o_widx, hidx, 8 * (nw + nh), hipMemcpyHostToDevice, s));
            k::gather(L1, reinterpret_cast<const uint32_t*>(A), reinterpret_cast<const uint64_t*>(A + J.o_widx), nw,
                      reinterpret_cast<uint32_t*>(A + J.o_wout), reinterpret_cast<const uint64_t*>(A + J.o_widx) + nw, nh, A + J.o_wout + wbytes);
            FR_HIP(ctx, hipMemcpyAsync(hp, A + J.o_wout, out_bytes, hipMemcpyDeviceToHost, s));
        }
        FR_HIP(ctx, hipStreamSynchronize(s));
        FR_HIP(ctx, hipGetLastError());
        ctx->phase_ms[3] = ms_since(J.t_start);  // gather done (second and last synchronise)
        for (uint32_t b = 0; b < count; b++) {
            words_of[b] = reinterpret_cast<const uint32_t*>(hp) + w_begin[b];
            hashes_of[b] = hp + wbytes + 32 * h_begin[b];
        }
    }

    // ---- assemble the Proofs (src/proof.rs:67-76) ----
    if (outs.size() != count) outs.assign(count, ProofData{});  // else: the caller's (recycled) objects, every field rewritten below
    for (uint32_t b = 0; b < count; b++) {
        ProveJob::Blob& bl = J.blobs[b];
        ProofData& out = outs[b];
        const LayerCounts* cnt = &counts[(size_t)b * (1 + n_inner)];
        const uint32_t* wv = words_of[b];
        const uint8_t* hv = hashes_of[b];
        size_t wi = 0, hi = 0;
        auto take_qm = [&]() {
            QM31 q{wv[wi], wv[wi + 1], wv[wi + 2], wv[wi + 3]};
            wi += 4;
            return q;
        };
        out.pcs_config = cfg;
        out.log_size_bound = J.sh.L;
        out.proof_of_work = bl.nonce;
        out.last_layer_poly = std::move(bl.lastv);
        out.evaluations.resize(n_evals[b]);
        for (auto& q : out.evaluations) q = take_qm();
        out.inner_layers.resize(n_inner);
        for (uint32_t li = 0; li <= n_inner; li++) {
            LayerProof& lp = li == 0 ? out.first_layer : out.inner_layers[li - 1];
            lp.commitment = bl.roots[li];
            lp.fri_witness.resize(cnt[li].n_witness);
            for (auto& q : lp.fri_witness) q = take_qm();
            lp.hash_witness.resize(cnt[li].n_hashes);
            if (cnt[li].n_hashes) memcpy(lp.hash_witness.data(), hv + 32 * hi, 32 * cnt[li].n_hashes);
            hi += cnt[li].n_hashes;
            lp.column_witness.clear();
        }
        memcpy(out_commitments + 32 * (size_t)b, bl.roots[0].data(), 32);
    }
    ctx->phase_ms[4] = ms_since(J.t_start);  // proofs assembled
    return FRIEDA_OK;
}

int prove(Ctx* ctx, const uint8_t* data, size_t len, bool data_on_device, const uint64_t* seed, frieda_pcs_config cfg,
          uint8_t out_commitment[32], ProofData& out) {
    int rc = prove_begin(ctx, data, len, data_on_device, seed, cfg);
    if (rc) return rc;
    return prove_finish(ctx, out_commitment, out);
}

}  // namespace frieda
