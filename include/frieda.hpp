// frieda.hpp — header-only C++17 mirror of frieda's public Rust API over the C ABI (frieda_hip.h).
//
//   frieda::api::commit(data, log_blowup_factor) -> Commitment                         (/root/reference/src/lib.rs:31)
//   frieda::api::generate_proof(data, seed, pcs_config) -> Proof                       (src/lib.rs:36)
//   frieda::proof::commit_and_generate_proof(data, seed, pcs_config) -> {Commitment, Proof}   (src/proof.rs:32)
//   frieda::api::verify(proof, seed) -> bool                                           (src/lib.rs:41)
//
// Same names, argument meaning and error behaviour: where the reference panics, frieda::Panic is thrown; verifier
// rejections return false.  Option<u64> is std::optional<uint64_t>.  A thread-local default context (device 0) backs the
// free functions; construct frieda::Context for other devices / streams.
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "frieda_hip.h"

namespace frieda {

using Commitment = std::array<uint8_t, 32>;  // src/commit.rs:9

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};
struct Panic : Error {  // the reference implementation panics here (FRIEDA_ERR_INVARIANT)
    using Error::Error;
};

inline void check(int status, const frieda_ctx* ctx = nullptr) {
    if (status == FRIEDA_OK) return;
    std::string msg = std::string("frieda_hip: ") + frieda_status_string(status);
    if (ctx) msg += std::string(": ") + frieda_last_error(ctx);
    if (status == FRIEDA_ERR_INVARIANT) throw Panic(status, msg);
    throw Error(status, msg);
}

// stwo FriConfig / PcsConfig as constructed at src/proof.rs:109-116
struct FriConfig {
    uint32_t log_blowup_factor = 4, log_last_layer_degree_bound = 0;
    size_t n_queries = 20;
};
struct PcsConfig {
    uint32_t pow_bits = 20;
    FriConfig fri_config;
    frieda_pcs_config c() const {
        return {pow_bits, fri_config.log_blowup_factor, fri_config.log_last_layer_degree_bound, (uint32_t)fri_config.n_queries};
    }
};

struct QM31 {
    uint32_t v[4];
    bool operator==(const QM31& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    bool operator!=(const QM31& o) const { return !(*this == o); }
};

// frieda::proof::Proof (src/proof.rs:19-26); the fields the reference's tests mutate are exposed as accessors
class Proof {
  public:
    Proof() = default;
    explicit Proof(frieda_proof* h) : h_(h) {}
    Proof(const Proof& o) { check(frieda_proof_clone(o.h_, &h_)); }
    Proof(Proof&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Proof& operator=(Proof o) {
        std::swap(h_, o.h_);
        return *this;
    }
    ~Proof() { frieda_proof_free(h_); }

    uint64_t proof_of_work() const { return frieda_proof_proof_of_work(h_); }
    void set_proof_of_work(uint64_t v) { frieda_proof_set_proof_of_work(h_, v); }
    uint32_t log_size_bound() const { return frieda_proof_log_size_bound(h_); }
    size_t n_inner_layers() const { return frieda_proof_n_inner_layers(h_); }
    Commitment first_layer_commitment() const {
        Commitment c;
        const uint8_t* p = frieda_proof_layer_commitment(h_, 0);
        for (int i = 0; i < 32; i++) c[i] = p[i];
        return c;
    }
    std::vector<QM31> evaluations() const {
        size_t n = frieda_proof_n_evaluations(h_);
        std::vector<QM31> out(n);
        const uint32_t* p = frieda_proof_evaluations(h_);
        for (size_t i = 0; i < n; i++)
            for (int c = 0; c < 4; c++) out[i].v[c] = p[4 * i + c];
        return out;
    }
    void set_evaluations(const std::vector<QM31>& ev) {
        check(frieda_proof_resize_evaluations(h_, ev.size()));
        uint32_t* p = frieda_proof_evaluations(h_);
        for (size_t i = 0; i < ev.size(); i++)
            for (int c = 0; c < 4; c++) p[4 * i + c] = ev[i].v[c];
    }
    std::vector<uint8_t> serialize() const {
        std::vector<uint8_t> b(frieda_proof_serialize(h_, nullptr, 0));
        frieda_proof_serialize(h_, b.data(), b.size());
        return b;
    }
    static Proof deserialize(const std::vector<uint8_t>& b) {
        frieda_proof* h = nullptr;
        check(frieda_proof_deserialize(b.data(), b.size(), &h));
        return Proof(h);
    }
    const frieda_proof* handle() const { return h_; }

  private:
    frieda_proof* h_ = nullptr;
};

class Context {
  public:
    explicit Context(int device = 0, void* stream = nullptr) { check(frieda_ctx_create(device, stream, &h_)); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ~Context() { frieda_ctx_destroy(h_); }

    // a tuning / A-B option of this context alone, named like the environment variable that sets its default ("FRIEDA_HOST_DECOMMIT", ...)
    void set_option(const char* name, int64_t value) { check(frieda_ctx_set_option(h_, name, value), h_); }

    Commitment commit(const uint8_t* data, size_t len, uint32_t log_blowup_factor) {
        Commitment root;
        check(frieda_commit(h_, data, len, log_blowup_factor, root.data()), h_);
        return root;
    }
    std::pair<Commitment, Proof> commit_and_generate_proof(const uint8_t* data, size_t len, std::optional<uint64_t> seed,
                                                           const PcsConfig& cfg) {
        Commitment root;
        frieda_proof* p = nullptr;
        uint64_t s = seed.value_or(0);
        check(frieda_commit_and_generate_proof(h_, data, len, seed ? &s : nullptr, cfg.c(), root.data(), &p), h_);
        return {root, Proof(p)};
    }
    // `count` equal-length blobs, blob i at data + i * stride: every kernel is launched once for the whole batch
    std::vector<Commitment> commit_batch(const uint8_t* data, size_t stride, size_t len, uint32_t count, uint32_t log_blowup_factor) {
        std::vector<Commitment> roots(count);
        if (count) check(frieda_commit_batch(h_, data, stride, len, count, log_blowup_factor, roots.data()->data()), h_);
        return roots;
    }
    std::vector<std::pair<Commitment, Proof>> commit_and_generate_proof_batch(const uint8_t* data, size_t stride, size_t len, uint32_t count,
                                                                              const uint64_t* seeds_or_null, const PcsConfig& cfg) {
        std::vector<std::pair<Commitment, Proof>> out;
        if (!count) return out;
        std::vector<Commitment> roots(count);
        std::vector<frieda_proof*> ps(count, nullptr);
        check(frieda_commit_and_generate_proof_batch(h_, data, stride, len, count, seeds_or_null, cfg.c(), roots.data()->data(), ps.data()), h_);
        out.reserve(count);
        for (uint32_t i = 0; i < count; i++) out.emplace_back(roots[i], Proof(ps[i]));
        return out;
    }
    // The reconstructor's half of the README's sampling flow (/root/reference/README.md:56-69): any >= 2^log_coef + 2 distinct
    // (position, value) pairs of the bit-reversed codeword — e.g. pooled from api::verify_samples of proofs with different seeds —
    // give the blob's `len` bytes back (frieda_reconstruct_points_device: no linear system, every pair checked against the result;
    // pairs that are not values of one polynomial throw Error).  Repeated positions are ignored (the first one counts).
    std::vector<uint8_t> reconstruct_from_samples(const std::vector<uint32_t>& positions, const std::vector<QM31>& values, uint32_t log_coef,
                                                  uint32_t log_domain, size_t len) {
        if (positions.size() != values.size() || positions.empty()) throw Error(FRIEDA_ERR_ARG, "reconstruct_from_samples: one value per position");
        std::vector<uint32_t> flat(4 * values.size());
        for (size_t i = 0; i < values.size(); i++)
            for (int c = 0; c < 4; c++) flat[4 * i + c] = values[i].v[c];  // [n_points][4 columns][1 entry]
        DeviceBuffer d_cells(h_, 4 * flat.size()), d_out(h_, len + 8);
        check(frieda_dev_upload(h_, d_cells.ptr, flat.data(), 4 * flat.size()), h_);
        check(frieda_reconstruct_points_device(h_, static_cast<const uint32_t*>(d_cells.ptr), positions.data(), (uint32_t)positions.size(), 0, log_coef,
                                               log_domain, len, d_out.ptr),
              h_);
        std::vector<uint8_t> out(len);
        if (len) check(frieda_dev_download(h_, out.data(), d_out.ptr, len), h_);
        return out;
    }
    frieda_ctx* handle() { return h_; }

  private:
    struct DeviceBuffer {  // frieda_dev_alloc / frieda_dev_free (Level B column storage), scoped
        frieda_ctx* ctx;
        void* ptr = nullptr;
        DeviceBuffer(frieda_ctx* c, size_t bytes) : ctx(c) { check(frieda_dev_alloc(c, bytes ? bytes : 1, &ptr), c); }
        DeviceBuffer(const DeviceBuffer&) = delete;
        DeviceBuffer& operator=(const DeviceBuffer&) = delete;
        ~DeviceBuffer() { frieda_dev_free(ctx, ptr); }
    };
    frieda_ctx* h_ = nullptr;
};

// A batch of independent blobs across the GPUs of one node (frieda_multi): blob i -> devices[i mod n]; roots gathered with RCCL.
class MultiContext {
  public:
    explicit MultiContext(const std::vector<int>& devices) { check(frieda_multi_create(devices.data(), (uint32_t)devices.size(), &h_)); }
    MultiContext(const MultiContext&) = delete;
    MultiContext& operator=(const MultiContext&) = delete;
    ~MultiContext() { frieda_multi_destroy(h_); }

    frieda_multi* handle() { return h_; }  // for the C entry points this class does not wrap (owned by the object)
    bool uses_rccl() const { return frieda_multi_uses_rccl(h_) != 0; }
    uint64_t gather_count() const { return frieda_multi_gather_count(h_); }
    uint32_t device_count() const { return frieda_multi_device_count(h_); }
    // a tuning option (frieda_ctx_set_option) on device slot d, e.g. the batch policy's "FRIEDA_BATCH_BUDGET_MB"
    void set_option(uint32_t device_slot, const char* name, int64_t value) {
        frieda_ctx* c = frieda_multi_ctx(h_, device_slot);
        if (!c) throw Error(FRIEDA_ERR_ARG, "no such device slot");
        check(frieda_ctx_set_option(c, name, value));
    }
    // hands the device workspaces the handle keeps between calls (two per device, up to the batch budget each) back; the next call allocates again
    void release_workspace() { mcheck(frieda_multi_release_workspace(h_)); }
    // the CPUs the worker thread of device slot d is pinned to (its GPU's NUMA node); empty: not pinned
    std::vector<int> near_cpus(uint32_t device_slot) const {
        std::vector<int> v(frieda_multi_near_cpus(h_, device_slot, nullptr, 0));
        if (!v.empty()) frieda_multi_near_cpus(h_, device_slot, v.data(), v.size());
        return v;
    }
    std::vector<Commitment> commit_many(const std::vector<std::vector<uint8_t>>& blobs, uint32_t log_blowup_factor) {
        std::vector<const uint8_t*> ptrs;
        std::vector<size_t> lens;
        for (const auto& b : blobs) ptrs.push_back(b.data()), lens.push_back(b.size());
        std::vector<Commitment> roots(blobs.size());
        if (!blobs.empty()) mcheck(frieda_commit_many(h_, ptrs.data(), lens.data(), (uint32_t)blobs.size(), log_blowup_factor, roots.data()->data()));
        return roots;
    }
    std::vector<std::pair<Commitment, Proof>> prove_many(const std::vector<std::vector<uint8_t>>& blobs, const uint64_t* seeds_or_null,
                                                         const PcsConfig& cfg) {
        std::vector<const uint8_t*> ptrs;
        std::vector<size_t> lens;
        for (const auto& b : blobs) ptrs.push_back(b.data()), lens.push_back(b.size());
        std::vector<std::pair<Commitment, Proof>> out;
        if (blobs.empty()) return out;
        std::vector<Commitment> roots(blobs.size());
        std::vector<frieda_proof*> ps(blobs.size(), nullptr);
        mcheck(frieda_prove_many(h_, ptrs.data(), lens.data(), (uint32_t)blobs.size(), seeds_or_null, cfg.c(), roots.data()->data(), ps.data()));
        for (size_t i = 0; i < blobs.size(); i++) out.emplace_back(roots[i], Proof(ps[i]));
        return out;
    }

  private:
    void mcheck(int status) {
        if (status == FRIEDA_OK) return;
        const std::string detail = frieda_multi_last_error(h_);
        if (status == FRIEDA_ERR_INVARIANT) throw Panic(status, detail);
        throw Error(status, detail);
    }
    frieda_multi* h_ = nullptr;
};

inline Context& default_context() {
    thread_local Context ctx(0);
    return ctx;
}

namespace proof {
inline std::pair<Commitment, Proof> commit_and_generate_proof(const std::vector<uint8_t>& data, std::optional<uint64_t> seed,
                                                              const PcsConfig& cfg) {
    return default_context().commit_and_generate_proof(data.data(), data.size(), seed, cfg);
}
}  // namespace proof

namespace api {
inline Commitment commit(const std::vector<uint8_t>& data, uint32_t log_blowup_factor) {
    return default_context().commit(data.data(), data.size(), log_blowup_factor);
}
inline Proof generate_proof(const std::vector<uint8_t>& data, std::optional<uint64_t> seed, const PcsConfig& cfg) {
    return proof::commit_and_generate_proof(data, seed, cfg).second;
}
inline bool verify(const Proof& proof, std::optional<uint64_t> seed) {
    int ok = 0;
    uint64_t s = seed.value_or(0);
    check(frieda_verify(proof.handle(), seed ? &s : nullptr, &ok));
    return ok != 0;
}
// verify + where the accepted proof sampled: evaluations()[i] sits at position [i] of the bit-reversed codeword; nullopt when the proof
// is rejected (the sampling client's half of the README's flow; pooled pairs feed frieda_reconstruct_points_device)
inline std::optional<std::vector<uint32_t>> verify_samples(const Proof& proof, std::optional<uint64_t> seed) {
    int ok = 0;
    uint64_t s = seed.value_or(0);
    std::vector<uint32_t> pos(proof.evaluations().size() + 1);
    size_t n = 0;
    check(frieda_verify_samples(proof.handle(), seed ? &s : nullptr, &ok, pos.data(), pos.size(), &n));
    if (!ok) return std::nullopt;
    pos.resize(n);
    return pos;
}
}  // namespace api

}  // namespace frieda
