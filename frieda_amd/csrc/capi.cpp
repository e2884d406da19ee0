// capi.cpp — the extern "C" shell of libfrieda_hip.so (declarations and reference citations: include/frieda_hip.h).
// No exception leaves this file: every entry point catches and maps to a status code (FR_GUARD_*, host.h).  The reconstruction entry
// points (frieda_circle_interpolate_cells / _cells_any / _points, frieda_reconstruct_cells_device / _points_device) live in reconstruct.cpp.
#include <string.h>

#include <algorithm>
#include <new>

#include "host.h"
#include "../../include/frieda_hip_testing.h"

using namespace frieda;

extern "C" {

uint32_t frieda_abi_version(void) { return FRIEDA_ABI_VERSION; }

const char* frieda_status_string(int status) {
    switch (status) {
        case FRIEDA_OK: return "ok";
        case FRIEDA_ERR_ARG: return "invalid argument";
        case FRIEDA_ERR_HIP: return "HIP runtime error";
        case FRIEDA_ERR_INVARIANT: return "invariant violated (the reference panics here)";
        case FRIEDA_ERR_NOMEM: return "out of memory";
        case FRIEDA_ERR_FORMAT: return "malformed proof image";
        default: return "unknown status";
    }
}

const char* frieda_last_error(const frieda_ctx* ctx) { return ctx ? ctx->c.err.c_str() : "null context"; }
const char* frieda_ctx_notes(const frieda_ctx* ctx) { return ctx ? ctx->c.notes.c_str() : ""; }

int frieda_ctx_create(int device, void* stream, frieda_ctx** out) {
    if (!out) return FRIEDA_ERR_ARG;
    *out = nullptr;
    frieda_ctx* ctx = new (std::nothrow) frieda_ctx();
    if (!ctx) return FRIEDA_ERR_NOMEM;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev) {
        delete ctx;
        return e != hipSuccess ? FRIEDA_ERR_HIP : FRIEDA_ERR_ARG;
    }
    ctx->c.device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return FRIEDA_ERR_HIP;
    }
    if (stream) {
        ctx->c.stream = reinterpret_cast<hipStream_t>(stream);
        ctx->c.own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return FRIEDA_ERR_HIP;
        }
        ctx->c.own_stream = true;
    }
    // Two kernels want more LDS than the 64 KB default.  A device (or build) that refuses either keeps every other path: the generic
    // transform kernel then takes 3 columns per workgroup (51 KB) and small domains take the general path.
    // (what gets degraded is recorded in `notes`, frieda_ctx_notes — a successful create leaves last_error empty)
    if (k::ntt_opt_in_dynamic_lds() != hipSuccess) {
        (void)hipGetLastError();
        ctx->c.tuning.lds_opt_in_ok = false;
        if (ctx->c.tuning.ntt_cpw > 3) ctx->c.tuning.ntt_cpw = 3;
        ctx->c.notes += "68 KB of dynamic LDS refused for the transform kernels: 3 columns per workgroup, no side-by-side fold kernel\n";
    }
    if (!k::small_first_opt_in()) {
        ctx->c.tuning.no_small_fused = true;
        ctx->c.notes += "the fused small-domain kernel does not fit this device's LDS: general path for small domains\n";
    }
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess)
        ctx->c.tuning.device_mem_bytes = mem_total;  // the batch policy's budget is a share of THIS device (batch_budget_bytes)
    else
        (void)hipGetLastError();
    *out = ctx;
    return FRIEDA_OK;
}

int frieda_ctx_destroy(frieda_ctx* ctx) {
    if (!ctx) return FRIEDA_ERR_ARG;
    delete ctx;
    return FRIEDA_OK;
}

int frieda_ctx_synchronize(frieda_ctx* ctx) {
    if (!ctx) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
}

int frieda_ctx_release_workspace(frieda_ctx* ctx) {
    if (!ctx) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    if (ctx->c.arena) FR_HIP(&ctx->c, hipFree(ctx->c.arena));
    ctx->c.arena = nullptr;
    ctx->c.arena_bytes = 0;
    if (ctx->c.pinned) FR_HIP(&ctx->c, hipHostFree(ctx->c.pinned));
    ctx->c.pinned = nullptr;
    ctx->c.pinned_bytes = 0;
    if (ctx->c.pinned_in) FR_HIP(&ctx->c, hipHostFree(ctx->c.pinned_in));
    ctx->c.pinned_in = nullptr;
    ctx->c.drop_twiddles();
    return FRIEDA_OK;
}

int frieda_ctx_set_twiddle_cache(frieda_ctx* ctx, int enabled) {
    if (!ctx) return FRIEDA_ERR_ARG;
    ctx->c.cache_twiddles = enabled != 0;
    return FRIEDA_OK;
}

int frieda_ctx_set_host_channel(frieda_ctx* ctx, int enabled) {
    if (!ctx) return FRIEDA_ERR_ARG;
    ctx->c.host_channel = enabled != 0;
    return FRIEDA_OK;
}

int frieda_ctx_last_prove_phases(const frieda_ctx* ctx, double out_ms[8]) {
    if (!ctx || !out_ms) return FRIEDA_ERR_ARG;
    for (int i = 0; i < 8; i++) out_ms[i] = ctx->c.phase_ms[i];
    return FRIEDA_OK;
}

int frieda_ctx_blake2s_ceiling(frieda_ctx* ctx, double* leaf_per_s, double* node_per_s) {
    if (!ctx) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(k::blake2s_ceiling_scratch_bytes());
    if (rc) return rc;
    if (k::blake2s_ceiling(ctx->c.stream, reinterpret_cast<uint32_t*>(ctx->c.arena), leaf_per_s, node_per_s))
        return ctx->c.fail(FRIEDA_ERR_HIP, "blake2s_ceiling: event timing failed");
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_ctx_blake2s_ceiling_ex(frieda_ctx* ctx, double out[6]) {
    if (!ctx || !out) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(k::blake2s_ceiling_scratch_bytes());
    if (rc) return rc;
    double clk[4] = {0, 0, 0, 0};
    if (k::blake2s_ceiling(ctx->c.stream, reinterpret_cast<uint32_t*>(ctx->c.arena), &out[0], &out[1], clk))
        return ctx->c.fail(FRIEDA_ERR_HIP, "blake2s_ceiling: event timing failed");
    FR_HIP(&ctx->c, hipGetLastError());
    out[2] = clk[0];
    out[3] = clk[1];
    out[4] = clk[2];
    out[5] = clk[3];
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_ctx_last_transcript(const frieda_ctx* ctx, uint32_t* n_layers, uint32_t* alphas, size_t cap_layers, uint8_t digest_before_grind[32]) {
    if (!ctx || !n_layers) return FRIEDA_ERR_ARG;
    const Ctx::LastTranscript& t = ctx->c.last_transcript;
    *n_layers = (uint32_t)t.alphas.size();
    if (alphas)
        for (size_t i = 0; i < t.alphas.size() && i < cap_layers; i++) memcpy(alphas + 4 * i, t.alphas[i].data(), 16);
    if (digest_before_grind)
        for (int w = 0; w < 8; w++)
            for (int b = 0; b < 4; b++) digest_before_grind[4 * w + b] = (uint8_t)(t.digest_before_grind[w] >> (8 * b));
    return FRIEDA_OK;
}

int frieda_ctx_set_option(frieda_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_NO_JOB(&ctx->c);
    if (!k::tuning_set(ctx->c.tuning, name, (long)value)) return ctx->c.fail(FRIEDA_ERR_ARG, std::string("unknown option or value out of range: ") + name);
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_ctx_test_set_draw_bound(frieda_ctx* ctx, uint32_t bound) {
    if (!ctx || bound > 2u * P31) return FRIEDA_ERR_ARG;
    ctx->c.test_draw_bound = bound ? bound : 2u * P31;
    return FRIEDA_OK;
}

int frieda_ctx_test_set_grind_first_log(frieda_ctx* ctx, uint32_t log_first) {
    if (!ctx || (log_first != 0 && (log_first < 8 || log_first > 40))) return FRIEDA_ERR_ARG;
    ctx->c.tuning.test_grind_first_log = log_first;
    return FRIEDA_OK;
}

int frieda_ctx_test_set_arena_limit(frieda_ctx* ctx, uint64_t bytes) {
    if (!ctx) return FRIEDA_ERR_ARG;
    ctx->c.tuning.test_arena_limit = bytes;
    return FRIEDA_OK;
}

size_t frieda_workspace_bytes(size_t len, uint32_t log_blowup_factor, uint32_t log_last_layer_degree_bound, int prove) {
    return workspace_bytes_per_blob(len, log_blowup_factor, log_last_layer_degree_bound, prove != 0, true);
}

int frieda_batch_plan(const frieda_ctx* ctx, size_t len, uint32_t log_blowup_factor, uint32_t log_last_layer_degree_bound, int prove,
                      uint32_t count, uint32_t in_flight, uint32_t* out_calls, size_t cap, uint32_t* n_calls) {
    if (!n_calls || in_flight == 0 || in_flight > 64) return FRIEDA_ERR_ARG;
    *n_calls = 0;
    try {
        const size_t ws = workspace_bytes_per_blob(len, log_blowup_factor, log_last_layer_degree_bound, prove != 0, true);
        if (!ws) return FRIEDA_ERR_ARG;
        const k::Tuning& t = ctx ? ctx->c.tuning : k::tuning_defaults();
        std::vector<uint32_t> calls;
        batch_cut(count, batch_per_call(t, ws, count, in_flight), in_flight, calls);
        *n_calls = (uint32_t)calls.size();
        if (calls.size() > cap) return out_calls ? FRIEDA_ERR_ARG : FRIEDA_OK;  // out_calls == NULL: a query for the count
        if (out_calls)
            for (size_t i = 0; i < calls.size(); i++) out_calls[i] = calls[i];
        return FRIEDA_OK;
    } catch (...) {
        return FRIEDA_ERR_NOMEM;
    }
}

int frieda_ctx_set_kernel_timing(frieda_ctx* ctx, int enabled) {
    if (!ctx) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    return ctx->c.set_kernel_timing(enabled != 0);
    FR_GUARD_END(ctx)
}

size_t frieda_ctx_kernel_timing_report(frieda_ctx* ctx, char* buf, size_t cap, int reset) {
    if (!ctx) return 0;
    try {
        (void)hipSetDevice(ctx->c.device);
        std::string r = ctx->c.kernel_timing_report(reset != 0);
        if (buf && cap > r.size()) memcpy(buf, r.c_str(), r.size() + 1);
        return r.size() + 1;
    } catch (...) {
        return 0;
    }
}

// ---- Level A ----
int frieda_commit(frieda_ctx* ctx, const uint8_t* data, size_t len, uint32_t log_blowup_factor, uint8_t out_root[32]) {
    if (!ctx || !out_root || (!data && len)) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return commit_host(&ctx->c, data, len, log_blowup_factor, out_root);
    FR_GUARD_END(ctx)
}

int frieda_commit_device(frieda_ctx* ctx, const void* d_data, size_t len, uint32_t log_blowup_factor, void* d_out_root) {
    if (!ctx || !d_out_root || (!d_data && len)) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return commit_device(&ctx->c, static_cast<const uint8_t*>(d_data), len, log_blowup_factor, static_cast<uint8_t*>(d_out_root), false);
    FR_GUARD_END(ctx)
}

// ---- ProofPool ----
static size_t proof_capacity_bytes(const ProofData& d) {
    size_t b = d.evaluations.capacity() * sizeof(QM31) + d.last_layer_poly.capacity() * sizeof(QM31);
    auto layer = [](const LayerProof& l) { return l.fri_witness.capacity() * sizeof(QM31) + l.hash_witness.capacity() * 32 + l.column_witness.capacity() * 4; };
    b += layer(d.first_layer);
    for (const LayerProof& l : d.inner_layers) b += layer(l);
    return b + d.inner_layers.capacity() * sizeof(LayerProof);
}
frieda_proof* ProofPool::get() {
    {
        std::lock_guard<std::mutex> g(mu);
        if (!free_list.empty()) {
            frieda_proof* p = free_list.back();
            free_list.pop_back();
            const size_t b = proof_capacity_bytes(p->p);
            bytes = bytes > b ? bytes - b : 0;
            return p;
        }
    }
    return new frieda_proof();
}
void ProofPool::put(frieda_proof* p) {
    if (!p) return;
    p->home.reset();
    const size_t b = proof_capacity_bytes(p->p);
    {
        std::lock_guard<std::mutex> g(mu);
        if (free_list.size() < MAX_ENTRIES && bytes + b <= MAX_BYTES) {
            free_list.push_back(p);
            bytes += b;
            return;
        }
    }
    delete p;
}
ProofPool::~ProofPool() {
    for (frieda_proof* p : free_list) delete p;
}
// a proof object for `ctx` to fill; on failure give it back with pool->put
static frieda_proof* proof_from_pool(frieda_ctx* ctx) {
    frieda_proof* p = ctx->pool->get();
    p->home = ctx->pool;
    return p;
}

static int prove_common(frieda_ctx* ctx, const void* data, size_t len, bool on_device, const uint64_t* seed, frieda_pcs_config cfg,
                        uint8_t* out_commitment, frieda_proof** out) {
    if (!ctx || !out || (!data && len)) return FRIEDA_ERR_ARG;
    *out = nullptr;
    FR_GUARD_BEGIN
    frieda_proof* p = proof_from_pool(ctx);
    uint8_t root[32];
    int rc = prove(&ctx->c, static_cast<const uint8_t*>(data), len, on_device, seed, cfg, root, p->p);
    if (rc != FRIEDA_OK) {
        ctx->pool->put(p);
        return rc;
    }
    if (out_commitment) memcpy(out_commitment, root, 32);
    *out = p;
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_commit_and_generate_proof(frieda_ctx* ctx, const uint8_t* data, size_t len, const uint64_t* seed, frieda_pcs_config cfg,
                                     uint8_t out_commitment[32], frieda_proof** out) {
    return prove_common(ctx, data, len, false, seed, cfg, out_commitment, out);
}
int frieda_commit_and_generate_proof_device(frieda_ctx* ctx, const void* d_data, size_t len, const uint64_t* seed,
                                            frieda_pcs_config cfg, uint8_t out_commitment[32], frieda_proof** out) {
    return prove_common(ctx, d_data, len, true, seed, cfg, out_commitment, out);
}
int frieda_generate_proof(frieda_ctx* ctx, const uint8_t* data, size_t len, const uint64_t* seed, frieda_pcs_config cfg,
                          frieda_proof** out) {
    return prove_common(ctx, data, len, false, seed, cfg, nullptr, out);
}

int frieda_prove_begin(frieda_ctx* ctx, const uint8_t* data, size_t len, const uint64_t* seed, frieda_pcs_config cfg) {
    if (!ctx || (!data && len)) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return prove_begin(&ctx->c, data, len, false, seed, cfg);
    FR_GUARD_END(ctx)
}
int frieda_prove_begin_device(frieda_ctx* ctx, const void* d_data, size_t len, const uint64_t* seed, frieda_pcs_config cfg) {
    if (!ctx || (!d_data && len)) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return prove_begin(&ctx->c, static_cast<const uint8_t*>(d_data), len, true, seed, cfg);
    FR_GUARD_END(ctx)
}
int frieda_prove_finish(frieda_ctx* ctx, uint8_t out_commitment[32], frieda_proof** out) {
    if (!ctx || !out) return FRIEDA_ERR_ARG;
    *out = nullptr;
    FR_GUARD_BEGIN
    frieda_proof* p = proof_from_pool(ctx);
    uint8_t root[32];
    int rc = prove_finish(&ctx->c, root, p->p);
    if (rc != FRIEDA_OK) {
        ctx->pool->put(p);
        return rc;
    }
    if (out_commitment) memcpy(out_commitment, root, 32);
    *out = p;
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

// ---- batches ----
namespace {
int batch_begin(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count, bool on_device, const uint64_t* seeds,
                frieda_pcs_config cfg) {
    if (!ctx || (!data && len) || count == 0) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return prove_begin_batch(&ctx->c, data, stride, len, count, on_device, seeds, cfg);
    FR_GUARD_END(ctx)
}
int batch_finish(frieda_ctx* ctx, uint32_t count, uint8_t* out_commitments, frieda_proof** out_proofs) {
    if (!ctx || !out_proofs || !out_commitments || count == 0) return FRIEDA_ERR_ARG;
    for (uint32_t i = 0; i < count; i++) out_proofs[i] = nullptr;
    FR_GUARD_BEGIN
    if (ctx->c.job && job_count(&ctx->c) != count) return ctx->c.fail(FRIEDA_ERR_ARG, "count differs from the batch in flight");
    // the proofs are assembled into recycled objects (their vectors keep their capacity)
    std::vector<frieda_proof*> objs(count, nullptr);
    std::vector<ProofData> outs(count);
    for (uint32_t i = 0; i < count; i++) {
        objs[i] = proof_from_pool(ctx);
        outs[i] = std::move(objs[i]->p);
    }
    int rc = prove_finish_batch(&ctx->c, out_commitments, outs);
    for (uint32_t i = 0; i < count; i++) {
        if (i < outs.size()) objs[i]->p = std::move(outs[i]);
        if (rc != FRIEDA_OK)
            ctx->pool->put(objs[i]);
        else
            out_proofs[i] = objs[i];
    }
    return rc;
    FR_GUARD_END(ctx)
}
}  // namespace

int frieda_prove_batch_begin(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count, const uint64_t* seeds,
                             frieda_pcs_config cfg) {
    return batch_begin(ctx, data, stride, len, count, false, seeds, cfg);
}
int frieda_prove_batch_begin_device(frieda_ctx* ctx, const void* d_data, size_t stride, size_t len, uint32_t count, const uint64_t* seeds,
                                    frieda_pcs_config cfg) {
    return batch_begin(ctx, static_cast<const uint8_t*>(d_data), stride, len, count, true, seeds, cfg);
}
int frieda_prove_batch_finish(frieda_ctx* ctx, uint32_t count, uint8_t* out_commitments, frieda_proof** out_proofs) {
    return batch_finish(ctx, count, out_commitments, out_proofs);
}

int frieda_commit_and_generate_proof_batch(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count,
                                           const uint64_t* seeds, frieda_pcs_config cfg, uint8_t* out_commitments,
                                           frieda_proof** out_proofs) {
    if (!out_proofs || !out_commitments) return FRIEDA_ERR_ARG;
    int rc = batch_begin(ctx, data, stride, len, count, false, seeds, cfg);
    return rc != FRIEDA_OK ? rc : batch_finish(ctx, count, out_commitments, out_proofs);
}

int frieda_commit_and_generate_proof_batch_device(frieda_ctx* ctx, const void* d_data, size_t stride, size_t len, uint32_t count,
                                                  const uint64_t* seeds, frieda_pcs_config cfg, uint8_t* out_commitments,
                                                  frieda_proof** out_proofs) {
    if (!out_proofs || !out_commitments) return FRIEDA_ERR_ARG;
    int rc = batch_begin(ctx, static_cast<const uint8_t*>(d_data), stride, len, count, true, seeds, cfg);
    return rc != FRIEDA_OK ? rc : batch_finish(ctx, count, out_commitments, out_proofs);
}

int frieda_commit_batch(frieda_ctx* ctx, const uint8_t* data, size_t stride, size_t len, uint32_t count, uint32_t log_blowup_factor,
                        uint8_t* out_roots) {
    if (!ctx || !out_roots || (!data && len) || count == 0) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return commit_batch(&ctx->c, data, stride, len, count, false, log_blowup_factor, out_roots);
    FR_GUARD_END(ctx)
}

int frieda_commit_batch_device(frieda_ctx* ctx, const void* d_data, size_t stride, size_t len, uint32_t count, uint32_t log_blowup_factor,
                               uint8_t* out_roots) {
    if (!ctx || !out_roots || (!d_data && len) || count == 0) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    return commit_batch(&ctx->c, static_cast<const uint8_t*>(d_data), stride, len, count, true, log_blowup_factor, out_roots);
    FR_GUARD_END(ctx)
}

int frieda_verify(const frieda_proof* proof, const uint64_t* seed, int* ok) {
    if (!proof || !ok) return FRIEDA_ERR_ARG;
    frieda_ctx* none = nullptr;
    FR_GUARD_BEGIN
    return verify(proof->p, seed, ok);
    FR_GUARD_END(none)
}

int frieda_verify_samples(const frieda_proof* proof, const uint64_t* seed, int* ok, uint32_t* out_positions, size_t cap, size_t* n_positions) {
    if (!proof || !ok || !n_positions || (cap && !out_positions)) return FRIEDA_ERR_ARG;
    frieda_ctx* none = nullptr;
    FR_GUARD_BEGIN
    *n_positions = 0;
    std::vector<uint32_t> q;
    const int rc = verify(proof->p, seed, ok, &q);
    if (rc != FRIEDA_OK || !*ok) return rc;
    if (q.size() != proof->p.evaluations.size()) return FRIEDA_ERR_INVARIANT;  // (an accepted proof has one evaluation per distinct query)
    *n_positions = q.size();
    if (cap < q.size()) return FRIEDA_ERR_ARG;
    if (!q.empty()) memcpy(out_positions, q.data(), 4 * q.size());
    return FRIEDA_OK;
    FR_GUARD_END(none)
}

// ---- Proof accessors ----
void frieda_proof_free(frieda_proof* p) {
    if (!p) return;
    if (std::shared_ptr<ProofPool> home = p->home)  // keeps the pool alive across put() even if this was its last reference
        home->put(p);
    else
        delete p;
}

int frieda_proof_clone(const frieda_proof* p, frieda_proof** out) {
    if (!p || !out) return FRIEDA_ERR_ARG;
    frieda_ctx* none = nullptr;
    FR_GUARD_BEGIN
    *out = new frieda_proof();
    (*out)->p = p->p;
    return FRIEDA_OK;
    FR_GUARD_END(none)
}
uint64_t frieda_proof_proof_of_work(const frieda_proof* p) { return p->p.proof_of_work; }
void frieda_proof_set_proof_of_work(frieda_proof* p, uint64_t nonce) { p->p.proof_of_work = nonce; }
frieda_pcs_config frieda_proof_pcs_config(const frieda_proof* p) { return p->p.pcs_config; }
uint32_t frieda_proof_log_size_bound(const frieda_proof* p) { return p->p.log_size_bound; }
size_t frieda_proof_n_evaluations(const frieda_proof* p) { return p->p.evaluations.size(); }
uint32_t* frieda_proof_evaluations(frieda_proof* p) { return reinterpret_cast<uint32_t*>(p->p.evaluations.data()); }
int frieda_proof_resize_evaluations(frieda_proof* p, size_t n) {
    if (!p) return FRIEDA_ERR_ARG;
    frieda_ctx* none = nullptr;
    FR_GUARD_BEGIN
    p->p.evaluations.resize(n, QM31{0, 0, 0, 0});
    return FRIEDA_OK;
    FR_GUARD_END(none)
}
size_t frieda_proof_n_inner_layers(const frieda_proof* p) { return p->p.inner_layers.size(); }
static const LayerProof* layer_of(const frieda_proof* p, size_t layer) {
    if (layer == 0) return &p->p.first_layer;
    if (layer - 1 < p->p.inner_layers.size()) return &p->p.inner_layers[layer - 1];
    return nullptr;
}
const uint8_t* frieda_proof_layer_commitment(const frieda_proof* p, size_t layer) {
    const LayerProof* l = layer_of(p, layer);
    return l ? l->commitment.data() : nullptr;
}
const uint32_t* frieda_proof_layer_fri_witness(const frieda_proof* p, size_t layer, size_t* n_qm31) {
    const LayerProof* l = layer_of(p, layer);
    if (n_qm31) *n_qm31 = l ? l->fri_witness.size() : 0;
    return l ? reinterpret_cast<const uint32_t*>(l->fri_witness.data()) : nullptr;
}
const uint8_t* frieda_proof_layer_hash_witness(const frieda_proof* p, size_t layer, size_t* n_hashes) {
    const LayerProof* l = layer_of(p, layer);
    if (n_hashes) *n_hashes = l ? l->hash_witness.size() : 0;
    return l ? reinterpret_cast<const uint8_t*>(l->hash_witness.data()) : nullptr;
}
const uint32_t* frieda_proof_layer_column_witness(const frieda_proof* p, size_t layer, size_t* n_m31) {
    const LayerProof* l = layer_of(p, layer);
    if (n_m31) *n_m31 = l ? l->column_witness.size() : 0;
    return l ? l->column_witness.data() : nullptr;
}
const uint32_t* frieda_proof_last_layer_poly(const frieda_proof* p, size_t* n_qm31) {
    if (n_qm31) *n_qm31 = p->p.last_layer_poly.size();
    return reinterpret_cast<const uint32_t*>(p->p.last_layer_poly.data());
}
size_t frieda_proof_serialize(const frieda_proof* p, uint8_t* buf, size_t cap) {
    if (!p) return 0;
    try {
        std::vector<uint8_t> b = serialize_proof(p->p);
        if (buf && cap >= b.size()) memcpy(buf, b.data(), b.size());
        return b.size();
    } catch (...) {
        return 0;
    }
}
int frieda_proof_deserialize(const uint8_t* buf, size_t len, frieda_proof** out) {
    if (!buf || !out) return FRIEDA_ERR_ARG;
    *out = nullptr;
    frieda_ctx* none = nullptr;
    FR_GUARD_BEGIN
    frieda_proof* p = new frieda_proof();
    if (!deserialize_proof(buf, len, p->p)) {
        delete p;
        return FRIEDA_ERR_FORMAT;
    }
    *out = p;
    return FRIEDA_OK;
    FR_GUARD_END(none)
}

// ---- Level B ----
int frieda_dev_alloc(frieda_ctx* ctx, size_t bytes, void** d_out) {
    if (!ctx || !d_out) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    hipError_t e = hipMalloc(d_out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        ctx->c.err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return FRIEDA_ERR_NOMEM;
    }
    return FRIEDA_OK;
}
int frieda_dev_free(frieda_ctx* ctx, void* d) {
    if (!ctx) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    FR_HIP(&ctx->c, hipFree(d));
    return FRIEDA_OK;
}
int frieda_dev_upload(frieda_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    if (bytes) FR_HIP(&ctx->c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->c.stream));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
}
int frieda_dev_download(frieda_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!h_dst || !d_src))) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    if (bytes) FR_HIP(&ctx->c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->c.stream));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
}

int frieda_dev_at(frieda_ctx* ctx, const uint32_t* d_col, size_t index, uint32_t* out) {
    if (!ctx || !d_col || !out) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    FR_HIP(&ctx->c, hipMemcpyAsync(out, d_col + index, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->c.stream));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
}
int frieda_dev_at_secure(frieda_ctx* ctx, const uint32_t* d_cols, size_t stride, size_t index, uint32_t out[4]) {
    if (!ctx || !d_cols || !out || index >= stride) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    // one strided copy: 4 rows of 4 bytes, `stride` words apart
    FR_HIP(&ctx->c, hipMemcpy2DAsync(out, sizeof(uint32_t), d_cols + index, sizeof(uint32_t) * stride, sizeof(uint32_t), 4,
                                     hipMemcpyDeviceToHost, ctx->c.stream));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
}
int frieda_bit_reverse_column(frieda_ctx* ctx, uint32_t* d_cols, size_t stride, uint32_t ncols, uint32_t log_size) {
    if (!ctx || !d_cols || ncols == 0 || ncols > 65535 || log_size > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    if (ncols > 1 && stride < ((size_t)1 << log_size)) return ctx->c.fail(FRIEDA_ERR_ARG, "column stride smaller than the column");
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    k::bit_reverse_columns(ctx->c.launch(), d_cols, stride, ncols, log_size);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
}

int frieda_codec_shape(size_t len, size_t* n_felts, size_t* n_padded, uint32_t* log_size) {
    CodecShape s = codec_shape(len);
    if (n_felts) *n_felts = s.n_felts;
    if (n_padded) *n_padded = s.n_padded;
    if (log_size) *log_size = s.log_size;
    return FRIEDA_OK;
}

int frieda_unpack30(frieda_ctx* ctx, const void* d_bytes, size_t len, uint32_t* d_coef, size_t n_out) {
    if (!ctx || !d_coef || (!d_bytes && len)) return FRIEDA_ERR_ARG;
    if (n_out < (8 * len + 29) / 30) return ctx->c.fail(FRIEDA_ERR_ARG, "n_out smaller than the felt count");
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    k::unpack30(ctx->c.launch(), static_cast<const uint8_t*>(d_bytes), len, d_coef, n_out);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
}

int frieda_precompute_twiddles(frieda_ctx* ctx, uint32_t log_domain, const uint32_t** d_twiddles, const uint32_t** d_inv_twiddles) {
    if (!ctx || log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    TwiddleSet ts;
    int rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    if (d_twiddles) *d_twiddles = ts.d_tw;
    if (d_inv_twiddles) *d_inv_twiddles = ts.d_itw;
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_circle_evaluate(frieda_ctx* ctx, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, uint32_t log_domain,
                           uint32_t* d_out) {
    if (!ctx || !d_coef || !d_out || ncols == 0 || ncols > 65535) return FRIEDA_ERR_ARG;
    if (log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN || log_coef > log_domain) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    TwiddleSet ts;
    int rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    k::circle_evaluate(ctx->c.launch(), d_coef, (size_t)1 << log_coef, ncols, log_coef, log_domain, ts.d_tw, ts.ds, d_out,
                       (size_t)1 << log_domain);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_circle_interpolate(frieda_ctx* ctx, const uint32_t* d_block, uint32_t ncols, uint32_t log_coef, uint32_t log_domain,
                              uint32_t block, uint32_t* d_coef) {
    if (!ctx || !d_block || !d_coef || ncols == 0 || ncols > 65535) return FRIEDA_ERR_ARG;
    if (log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN || log_coef > log_domain) return FRIEDA_ERR_ARG;
    if ((uint64_t)block >= ((uint64_t)1 << (log_domain - log_coef))) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    TwiddleSet ts;
    int rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    k::circle_interpolate_block(ctx->c.launch(), d_block, (size_t)1 << log_coef, ncols, log_coef, log_domain, block, ts.d_itw, ts.ds,
                                d_coef, (size_t)1 << log_coef);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_pack30(frieda_ctx* ctx, const uint32_t* d_felts, size_t n_felts, void* d_bytes, size_t len) {
    if (!ctx || (len && (!d_felts || !d_bytes))) return FRIEDA_ERR_ARG;
    if ((8 * len + 29) / 30 > n_felts) return ctx->c.fail(FRIEDA_ERR_ARG, "len needs more felts than given");
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    k::pack30(ctx->c.launch(), d_felts, n_felts, static_cast<uint8_t*>(d_bytes), len);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
}

int frieda_reconstruct_device(frieda_ctx* ctx, const uint32_t* d_block, uint32_t log_coef, uint32_t log_domain, uint32_t block,
                              size_t len, void* d_out_bytes) {
    if (!ctx || !d_block || (len && !d_out_bytes)) return FRIEDA_ERR_ARG;
    if (log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN || log_coef > log_domain) return FRIEDA_ERR_ARG;
    if ((uint64_t)block >= ((uint64_t)1 << (log_domain - log_coef))) return FRIEDA_ERR_ARG;
    const size_t n_felts = (size_t)4 << log_coef;
    if ((8 * len + 29) / 30 > n_felts) return ctx->c.fail(FRIEDA_ERR_ARG, "len does not fit the polynomial");
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(sizeof(uint32_t) * n_felts);
    if (rc) return rc;
    TwiddleSet ts;
    rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    uint32_t* coef = reinterpret_cast<uint32_t*>(ctx->c.arena);
    k::circle_interpolate_block(ctx->c.launch(), d_block, (size_t)1 << log_coef, 4, log_coef, log_domain, block, ts.d_itw, ts.ds, coef,
                                (size_t)1 << log_coef);
    k::pack30(ctx->c.launch(), coef, n_felts, static_cast<uint8_t*>(d_out_bytes), len);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_merkle_commit_layer(frieda_ctx* ctx, uint32_t log_size, const void* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                               void* d_out) {
    if (!ctx || !d_out || log_size > FRIEDA_MAX_LOG_DOMAIN || (ncols && !d_cols) || ncols > 1024) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    hipStream_t s = ctx->c.stream;
    const size_t n = (size_t)1 << log_size;
    uint8_t* out = static_cast<uint8_t*>(d_out);
    const uint8_t* prev = static_cast<const uint8_t*>(d_prev);
    if (!prev && ncols == 4) {
        k::merkle_leaf4(ctx->c.launch(), d_cols[0], d_cols[1], d_cols[2], d_cols[3], n, out);
    } else if (prev && ncols == 0) {
        k::merkle_node(ctx->c.launch(), prev, n, out);
    } else {
        // general shape: the column pointer table goes through the workspace arena
        int rc = ctx->c.ensure_arena(sizeof(void*) * (ncols ? ncols : 1));
        if (rc) return rc;
        if (ncols) FR_HIP(&ctx->c, hipMemcpyAsync(ctx->c.arena, d_cols, sizeof(void*) * ncols, hipMemcpyHostToDevice, s));
        k::merkle_layer_generic(ctx->c.launch(), prev, reinterpret_cast<const uint32_t* const*>(ctx->c.arena), ncols, n, out);
        FR_HIP(&ctx->c, hipStreamSynchronize(s));  // d_cols is caller memory; the arena slot is reused by the next call
    }
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

size_t frieda_merkle_layer_offset(uint32_t log_size, uint32_t layer_log) { return k::merkle_layer_offset(log_size, layer_log); }

int frieda_merkle_commit(frieda_ctx* ctx, const uint32_t* d_cols, uint32_t log_size, void* d_layers) {
    if (!ctx || !d_cols || !d_layers || log_size > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    const size_t n = (size_t)1 << log_size;
    k::merkle_tree4(ctx->c.launch(), d_cols, d_cols + n, d_cols + 2 * n, d_cols + 3 * n, log_size, static_cast<uint8_t*>(d_layers));
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
}

int frieda_merkle_root(frieda_ctx* ctx, const uint32_t* d_cols, uint32_t log_size, void* d_root) {
    if (!ctx || !d_cols || !d_root || log_size > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(k::merkle_root_scratch_bytes(log_size));
    if (rc) return rc;
    const size_t n = (size_t)1 << log_size;
    k::merkle_root4(ctx->c.launch(), d_cols, d_cols + n, d_cols + 2 * n, d_cols + 3 * n, log_size, ctx->c.arena, static_cast<uint8_t*>(d_root));
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_fold_circle_into_line(frieda_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, uint32_t log_domain, const uint32_t alpha[4]) {
    if (!ctx || !d_dst || !d_src || !alpha || log_domain < 1 || log_domain > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    TwiddleSet ts;
    int rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    k::fold_circle_into_line(ctx->c.launch(), d_dst, (size_t)1 << (log_domain - 1), d_src, (size_t)1 << log_domain, log_domain, ts.d_itw,
                             ts.ds, k::Alpha{{alpha[0], alpha[1], alpha[2], alpha[3]}});
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_circle_evaluate_fold2(frieda_ctx* ctx, const uint32_t* d_coeffs, uint32_t log_size, uint32_t log_domain, uint32_t* d_evals,
                                 const uint32_t alpha0[4], int accumulate_line1, uint32_t* d_line1, const uint32_t alpha1[4], uint32_t* d_line2) {
    if (!ctx || !d_coeffs || !d_evals || !alpha0 || !alpha1 || !d_line1 || !d_line2 || log_domain < 2 || log_domain > FRIEDA_MAX_LOG_DOMAIN ||
        log_size > log_domain)
        return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    for (int i = 0; i < 4; i++)
        if (alpha0[i] >= P31 || alpha1[i] >= P31) return ctx->c.fail(FRIEDA_ERR_ARG, "alpha coordinates must be canonical M31 values");
    {  // the fused pass writes the lines while other workgroups still write the evaluation: no two of the four buffers may overlap
        const uintptr_t b[4] = {reinterpret_cast<uintptr_t>(d_coeffs), reinterpret_cast<uintptr_t>(d_evals), reinterpret_cast<uintptr_t>(d_line1),
                                reinterpret_cast<uintptr_t>(d_line2)};
        const size_t len[4] = {(size_t)16 << log_size, (size_t)16 << log_domain, (size_t)16 << (log_domain - 1), (size_t)16 << (log_domain - 2)};
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++)
                if (b[i] < b[j] + len[j] && b[j] < b[i] + len[i]) return ctx->c.fail(FRIEDA_ERR_ARG, "coefficient, evaluation and line buffers must not overlap");
    }
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    TwiddleSet ts;
    int rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    k::EncodeFoldSink fs{};
    fs.itw = ts.d_itw;
    for (int i = 0; i < 4; i++) fs.alpha0[i] = alpha0[i], fs.alpha1[i] = alpha1[i];
    fs.accumulate = accumulate_line1 != 0;
    fs.line1 = d_line1;
    fs.line2 = d_line2;
    hipError_t fe = hipSuccess;
    (void)k::circle_evaluate_fold2(ctx->c.launch(), d_coeffs, (size_t)1 << log_size, log_size, log_domain, ts.d_tw, ts.ds, d_evals,
                                   (size_t)1 << log_domain, fs, &fe);
    FR_HIP(&ctx->c, fe);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_fold_line(frieda_ctx* ctx, const uint32_t* d_src, uint32_t line_log, uint32_t log_domain, const uint32_t alpha[4],
                     uint32_t* d_dst) {
    if (!ctx || !d_dst || !d_src || !alpha || line_log < 1 || log_domain < 2 || line_log > log_domain - 1 ||
        log_domain > FRIEDA_MAX_LOG_DOMAIN)
        return FRIEDA_ERR_ARG;
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    TwiddleSet ts;
    int rc = ctx->c.get_twiddles(log_domain, ts);
    if (rc) return rc;
    k::fold_line(ctx->c.launch(), d_src, (size_t)1 << line_log, line_log, log_domain, ts.d_itw, ts.ds,
                 k::Alpha{{alpha[0], alpha[1], alpha[2], alpha[3]}}, d_dst, (size_t)1 << (line_log - 1));
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_circle_extend(frieda_ctx* ctx, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, uint32_t log_size, uint32_t* d_out) {
    if (!ctx || !d_coef || !d_out || ncols == 0 || ncols > 65535 || log_size > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    if (log_size < log_coef) return ctx->c.fail(FRIEDA_ERR_INVARIANT, "extend: log_size smaller than the polynomial's (stwo asserts log_size >= poly.log_size())");
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    k::circle_extend(ctx->c.launch(), d_coef, ncols, log_coef, log_size, d_out);
    FR_HIP(&ctx->c, hipGetLastError());
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_circle_eval_at_point(frieda_ctx* ctx, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, const uint32_t point_x[4],
                                const uint32_t point_y[4], uint32_t* out) {
    if (!ctx || !d_coef || !point_x || !point_y || !out || ncols == 0 || ncols > 65535 || log_coef > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    for (int i = 0; i < 4; i++)
        if (point_x[i] >= P31 || point_y[i] >= P31) return ctx->c.fail(FRIEDA_ERR_ARG, "eval_at_point: point coordinates must be canonical M31 words");
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(k::eval_at_point_scratch_bytes(ncols, log_coef));
    if (rc) return rc;
    // CpuBackend::eval_at_point: mappings = [y, x, double_x(x), ...]: the factor of coefficient-index bit b
    k::EvalFactors f{};
    const QM31 one{1, 0, 0, 0};
    f.f[0] = QM31{point_y[0], point_y[1], point_y[2], point_y[3]};
    QM31 x{point_x[0], point_x[1], point_x[2], point_x[3]};
    for (uint32_t b = 1; b < 32; b++) {
        f.f[b] = x;
        const QM31 sq = qm_mul(x, x);
        x = qm_sub(qm_add(sq, sq), one);
    }
    const uint32_t* d_res = k::circle_eval_at_point(ctx->c.launch(), d_coef, ncols, log_coef, f, reinterpret_cast<uint32_t*>(ctx->c.arena));
    FR_HIP(&ctx->c, hipGetLastError());
    FR_HIP(&ctx->c, hipMemcpyAsync(out, d_res, 16 * (size_t)ncols, hipMemcpyDeviceToHost, ctx->c.stream));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_fri_decompose(frieda_ctx* ctx, const uint32_t* d_eval, uint32_t log_size, uint32_t* d_g, uint32_t out_lambda[4]) {
    if (!ctx || !d_eval || !d_g || !out_lambda || log_size > FRIEDA_MAX_LOG_DOMAIN) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(k::decompose_scratch_bytes(log_size));
    if (rc) return rc;
    uint32_t* d_scratch = reinterpret_cast<uint32_t*>(ctx->c.arena);
    k::fri_decompose(ctx->c.launch(), d_eval, log_size, d_g, d_scratch);
    FR_HIP(&ctx->c, hipGetLastError());
    FR_HIP(&ctx->c, hipMemcpyAsync(out_lambda, d_scratch, 16, hipMemcpyDeviceToHost, ctx->c.stream));
    FR_HIP(&ctx->c, hipStreamSynchronize(ctx->c.stream));
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

int frieda_grind(frieda_ctx* ctx, const uint8_t digest[32], uint32_t pow_bits, uint64_t* nonce) {
    if (!ctx || !digest || !nonce || pow_bits > 48) return FRIEDA_ERR_ARG;
    FR_NO_JOB(&ctx->c);
    FR_GUARD_BEGIN
    FR_HIP(&ctx->c, hipSetDevice(ctx->c.device));
    int rc = ctx->c.ensure_arena(256);
    if (rc) return rc;
    hipStream_t s = ctx->c.stream;
    uint32_t dw[8];
    for (int i = 0; i < 8; i++)
        dw[i] = (uint32_t)digest[4 * i] | ((uint32_t)digest[4 * i + 1] << 8) | ((uint32_t)digest[4 * i + 2] << 16) | ((uint32_t)digest[4 * i + 3] << 24);
    unsigned long long* d_res = reinterpret_cast<unsigned long long*>(ctx->c.arena);
    FR_HIP(&ctx->c, hipMemsetAsync(d_res, 0xFF, 8, s));
    uint64_t base = 0, chunk = (uint64_t)1 << 22, found = ~0ull;
    for (;;) {
        k::grind_scan(ctx->c.launch(), dw, pow_bits, base, chunk, d_res);
        FR_HIP(&ctx->c, hipMemcpyAsync(&found, d_res, 8, hipMemcpyDeviceToHost, s));
        FR_HIP(&ctx->c, hipStreamSynchronize(s));
        if (found != ~0ull) break;
        base += chunk;
        if (chunk < ((uint64_t)1 << 28)) chunk <<= 1;
    }
    *nonce = found;
    return FRIEDA_OK;
    FR_GUARD_END(ctx)
}

}  // extern "C"
