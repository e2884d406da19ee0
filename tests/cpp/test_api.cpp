// test_api.cpp — the reference's own unit tests (/root/reference/src/commit.rs:28-38, src/proof.rs:119-193,
// src/lib.rs:52-85), transcribed to C++ over include/frieda.hpp.  Built by __graft_entry__.build(), run on the GPU box by
// tests/test_gpu_parity.py::test_cpp_api_harness.  argv[1] = path of the `blob` fixture.  Exit code 0 = all passed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

#include <sched.h>

#include "frieda.hpp"
#include "frieda_hip_testing.h"

using namespace frieda;

static int failures = 0;
#define CHECK(cond)                                                  \
    do {                                                             \
        if (!(cond)) {                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                              \
        }                                                            \
    } while (0)

static const PcsConfig PCS_CONFIG{20, FriConfig{4, 1, 20}};  // src/proof.rs:109-116

// `test_api.bin <blob> multi <n_slots>`: the multi-GPU entry points over n_slots device slots, all of them device 0 (the test
// box has one GPU).  n_slots == 1 takes the no-exchange path unless FRIEDA_MULTI_FORCE_RCCL=1 (then the real one-rank RCCL
// collective runs); n_slots > 1 needs FRIEDA_RCCL_PATH = tests/cpp/librccl_stub.so (real RCCL refuses a device listed twice).
static int multi_mode(const std::vector<uint8_t>& blob, int n_slots, bool distinct_devices = false) {
    std::vector<std::vector<uint8_t>> blobs;
    blobs.push_back(blob);  // the reference's fixture: its root is the golden root
    for (int i = 0; i < 6; i++) {  // ragged lengths, an empty blob among them
        std::vector<uint8_t> d(i == 3 ? 0 : 700 + 1311 * i);
        for (size_t j = 0; j < d.size(); j++) d[j] = (uint8_t)(j * 29 + i * 5 + (j >> 5));
        blobs.push_back(d);
    }
    // a run of equal lengths: each device takes its share through the batched kernels, cut into calls by the library's batch policy
    // (11 blobs on 1 - 3 slots; 8 slots get 8 x 9 + 3 so that every slot sees a multi-call run, one slot a longer one)
    const int run_len = n_slots >= 4 ? 9 * n_slots + 3 : 11;
    for (int i = 0; i < run_len; i++) {
        std::vector<uint8_t> d(3000);
        for (size_t j = 0; j < d.size(); j++) d[j] = (uint8_t)(j * 13 + i * 31 + (j >> 7));
        blobs.push_back(d);
    }
    const uint8_t golden[32] = {209, 162, 213, 6,  157, 197, 135, 229, 93,  194, 156, 198, 37, 90, 249, 55,
                                255, 127, 237, 14, 228, 27,  223, 90,  249, 135, 23,  249, 215, 79, 96,  232};
    // distinct_devices: slots 0 .. n_slots-1 are devices 0 .. n_slots-1 (a real multi-GPU node, real RCCL with n > 1)
    std::vector<int> devs(n_slots, 0);
    if (distinct_devices)
        for (int i = 0; i < n_slots; i++) devs[i] = i;
    MultiContext mc(devs);
    const bool want_rccl = n_slots > 1 || (getenv("FRIEDA_MULTI_FORCE_RCCL") && getenv("FRIEDA_MULTI_FORCE_RCCL")[0] == '1');
    CHECK(mc.uses_rccl() == want_rccl);
    auto roots = mc.commit_many(blobs, 4);
    CHECK(roots.size() == blobs.size());
    CHECK(std::memcmp(roots[0].data(), golden, 32) == 0);
    for (size_t i = 0; i < blobs.size(); i++) CHECK(roots[i] == api::commit(blobs[i], 4));
    std::vector<uint64_t> seeds;
    std::vector<std::vector<uint8_t>> provable;  // the empty blob is too small for the FRI configuration
    for (size_t i = 0; i < blobs.size(); i++)
        if (!blobs[i].empty()) provable.push_back(blobs[i]), seeds.push_back(1000 + i);
    auto proofs = mc.prove_many(provable, seeds.data(), PCS_CONFIG);
    CHECK(proofs.size() == provable.size());
    for (size_t i = 0; i < proofs.size(); i++) {
        auto single = proof::commit_and_generate_proof(provable[i], seeds[i], PCS_CONFIG);
        CHECK(proofs[i].first == single.first);
        CHECK(proofs[i].second.serialize() == single.second.serialize());
        CHECK(api::verify(proofs[i].second, seeds[i]));
    }
    auto unseeded = mc.prove_many(provable, nullptr, PCS_CONFIG);
    for (size_t i = 0; i < unseeded.size(); i++) CHECK(api::verify(unseeded[i].second, std::nullopt) && unseeded[i].first == proofs[i].first);
    CHECK(mc.gather_count() == (want_rccl ? 3u : 0u));
    // a batch with a blob the reference panics on: the whole call reports it and hands out no proof
    bool panicked = false;
    try {
        mc.prove_many(blobs, nullptr, PCS_CONFIG);
    } catch (const Panic&) {
        panicked = true;
    }
    CHECK(panicked);
    CHECK(mc.commit_many({}, 4).empty());
    // and the handle is usable afterwards
    CHECK(mc.commit_many(blobs, 4) == roots);
    {  // the batch policy is a performance knob, never a result: a 1 MB budget (one blob per call) and three calls per context
       // on every slot give the same roots and proofs
        CHECK(mc.device_count() == (uint32_t)n_slots);
        for (int d = 0; d < n_slots; d++) mc.set_option(d, "FRIEDA_BATCH_BUDGET_MB", 1);
        CHECK(mc.commit_many(blobs, 4) == roots);
        auto again = mc.prove_many(provable, seeds.data(), PCS_CONFIG);
        for (size_t i = 0; i < again.size(); i++) CHECK(again[i].first == proofs[i].first && again[i].second.serialize() == proofs[i].second.serialize());
        for (int d = 0; d < n_slots; d++) mc.set_option(d, "FRIEDA_BATCH_BUDGET_MB", 0), mc.set_option(d, "FRIEDA_BATCH_CALLS_PER_CTX", 3);
        CHECK(mc.commit_many(blobs, 4) == roots);
        again = mc.prove_many(provable, seeds.data(), PCS_CONFIG);
        for (size_t i = 0; i < again.size(); i++) CHECK(again[i].first == proofs[i].first && again[i].second.serialize() == proofs[i].second.serialize());
        mc.release_workspace();  // the handle stays usable: the next call allocates again
        CHECK(mc.commit_many(blobs, 4) == roots);
        // every slot short of memory at once: a test limit of 1.5 blobs' workspace on each slot's contexts makes every worker's
        // first cut of the equal-length run (3 or more blobs per call) fail with FRIEDA_ERR_NOMEM; the workers retry with halved calls,
        // concurrently, and the results do not move.  Below one blob's workspace the call fails and names a device.
        for (int d = 0; d < n_slots; d++) mc.set_option(d, "FRIEDA_BATCH_CALLS_PER_CTX", 1);
        std::vector<std::vector<uint8_t>> run_blobs(blobs.end() - run_len, blobs.end());
        std::vector<Commitment> run_roots(roots.end() - run_len, roots.end());
        std::vector<uint64_t> run_seeds(seeds.end() - run_len, seeds.end());
        const size_t ws_p = frieda_workspace_bytes(3000, 4, 0, 1), ws_c = frieda_workspace_bytes(3000, 4, 0, 0);
        CHECK(ws_p > ws_c && ws_c > 0);
        for (int d = 0; d < n_slots; d++) CHECK(frieda_ctx_test_set_arena_limit(frieda_multi_ctx(mc.handle(), d), ws_p + ws_p / 2) == FRIEDA_OK);
        auto tight = mc.prove_many(run_blobs, run_seeds.data(), PCS_CONFIG);
        CHECK(tight.size() == (size_t)run_len);
        for (int i = 0; i < run_len; i++) {
            const auto& want = proofs[proofs.size() - run_len + i];
            CHECK(tight[i].first == want.first && tight[i].second.serialize() == want.second.serialize());
        }
        for (int d = 0; d < n_slots; d++) CHECK(frieda_ctx_test_set_arena_limit(frieda_multi_ctx(mc.handle(), d), ws_c + ws_c / 2) == FRIEDA_OK);
        CHECK(mc.commit_many(run_blobs, 4) == run_roots);
        for (int d = 0; d < n_slots; d++) CHECK(frieda_ctx_test_set_arena_limit(frieda_multi_ctx(mc.handle(), d), ws_c / 2) == FRIEDA_OK);
        bool nomem = false;
        try {
            mc.commit_many(run_blobs, 4);
        } catch (const Error& e) {
            nomem = e.status == FRIEDA_ERR_NOMEM && std::string(e.what()).find("device") != std::string::npos;
        }
        CHECK(nomem);
        for (int d = 0; d < n_slots; d++) CHECK(frieda_ctx_test_set_arena_limit(frieda_multi_ctx(mc.handle(), d), 0) == FRIEDA_OK);
        CHECK(mc.commit_many(blobs, 4) == roots);
        bool refused = false;
        try {
            mc.set_option(0, "FRIEDA_BATCH_CALLS_PER_CTX", 0);
        } catch (const Error&) {
            refused = true;
        }
        CHECK(refused);
        // NUMA placement: whatever the platform reports, the list is a subset of the CPUs this process may run on
        cpu_set_t have;
        CPU_ZERO(&have);
        CHECK(sched_getaffinity(0, sizeof(have), &have) == 0);
        for (int d = 0; d < n_slots; d++)
            for (int c : mc.near_cpus(d)) CHECK(c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &have));
        // two workers either sit on the same NUMA node (the same CPU list) or on different ones (disjoint lists): never a partial overlap,
        // and never one list for devices the platform puts on different nodes
        for (int a = 0; a < n_slots; a++)
            for (int b = a + 1; b < n_slots; b++) {
                const std::vector<int> ca = mc.near_cpus(a), cb = mc.near_cpus(b);
                size_t common = 0;
                for (int c : ca) common += std::count(cb.begin(), cb.end(), c);
                CHECK(common == 0 || (ca == cb));
            }
        std::printf("near_cpus(slot 0): %zu\n", mc.near_cpus(0).size());
    }
    std::printf("multi n_slots=%d rccl=%d gathers=%llu: %s (%d failures)\n", n_slots, (int)mc.uses_rccl(), (unsigned long long)mc.gather_count(),
                failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    CHECK(data.size() == 262146);
    if (argc >= 4 && std::strcmp(argv[2], "multi") == 0) return multi_mode(data, std::atoi(argv[3]));
    if (argc >= 4 && std::strcmp(argv[2], "multi_real") == 0) return multi_mode(data, std::atoi(argv[3]), true);

    // test_commit (src/commit.rs:28-38): the golden root
    const uint8_t golden[32] = {209, 162, 213, 6,  157, 197, 135, 229, 93,  194, 156, 198, 37, 90, 249, 55,
                                255, 127, 237, 14, 228, 27,  223, 90,  249, 135, 23,  249, 215, 79, 96,  232};
    Commitment root = api::commit(data, 4);
    CHECK(std::memcmp(root.data(), golden, 32) == 0);

    // test_generate_proof / test_commit_and_generate_proof (src/proof.rs:119-135)
    auto [commitment, proof] = proof::commit_and_generate_proof(data, std::nullopt, PCS_CONFIG);
    CHECK(proof.n_inner_layers() != 0);
    CHECK(commitment == root);
    CHECK(proof.first_layer_commitment() == commitment);
    // test_verify_proof
    CHECK(api::verify(proof, std::nullopt));
    {  // the same verification also says where the proof sampled: one ascending position per evaluation, none for a rejected proof
        auto pos = api::verify_samples(proof, std::nullopt);
        CHECK(pos.has_value() && pos->size() == proof.evaluations().size() && std::is_sorted(pos->begin(), pos->end()));
        CHECK(!api::verify_samples(proof, 5).has_value());
    }
    {  // invalid pow
        Proof p = proof;
        p.set_proof_of_work(p.proof_of_work() + 1);
        CHECK(!api::verify(p, std::nullopt));
    }
    {  // invalid evaluations: evaluations[0] += (1, 1, 1, 1)
        Proof p = proof;
        auto ev = p.evaluations();
        for (int c = 0; c < 4; c++) ev[0].v[c] = (ev[0].v[c] + 1) % 0x7fffffffu;
        p.set_evaluations(ev);
        CHECK(!api::verify(p, std::nullopt));
    }
    {  // reversed
        Proof p = proof;
        auto ev = p.evaluations();
        std::reverse(ev.begin(), ev.end());
        p.set_evaluations(ev);
        CHECK(!api::verify(p, std::nullopt));
    }
    {  // popped: #[should_panic]
        Proof p = proof;
        auto ev = p.evaluations();
        ev.pop_back();
        p.set_evaluations(ev);
        bool panicked = false;
        try {
            api::verify(p, std::nullopt);
        } catch (const Panic&) {
            panicked = true;
        }
        CHECK(panicked);
    }
    {  // swap(0, 1)
        Proof p = proof;
        auto ev = p.evaluations();
        std::swap(ev[0], ev[1]);
        p.set_evaluations(ev);
        CHECK(!api::verify(p, std::nullopt));
    }
    {  // seeds (src/proof.rs:183-193)
        Proof p1 = api::generate_proof(data, 1, PCS_CONFIG), p2 = api::generate_proof(data, 2, PCS_CONFIG);
        CHECK(p1.evaluations() != p2.evaluations());
        CHECK(api::verify(p1, 1) && api::verify(p2, 2));
        CHECK(!api::verify(p1, 2) && !api::verify(p2, 1));
    }
    {  // test_end_to_end (src/lib.rs:52-85)
        const char* s = "This is the original data that needs to be made available.";
        std::vector<uint8_t> d(s, s + std::strlen(s));
        Commitment c = api::commit(d, 4);
        Proof p = api::generate_proof(d, std::nullopt, PcsConfig{20, FriConfig{4, 0, 20}});
        CHECK(api::verify(p, std::nullopt));
        CHECK(p.first_layer_commitment() == c);
        Proof q = Proof::deserialize(p.serialize());
        CHECK(api::verify(q, std::nullopt));
    }
    {  // the README's sampling loop (/root/reference/README.md:56-69) with what src/ has: a sample is a proof.  Clients with different
       // seeds each verify theirs and learn where it sampled; the pooled (position, value) pairs rebuild the data once there are
       // 2^L + 2 distinct ones; a forged pair in the pool is reported, never answered with wrong bytes.
        std::vector<uint8_t> d(1024);
        for (size_t i = 0; i < d.size(); i++) d[i] = (uint8_t)(i % 256);  // benches/commit.rs:6 -> 274 felts -> 2^7 coefficients per column
        const uint32_t log_coef = 7, log_domain = 11;
        const PcsConfig cfg{8, FriConfig{4, 0, 40}};
        std::vector<uint32_t> positions;
        std::vector<QM31> values;
        std::vector<bool> have((size_t)1 << log_domain, false);
        size_t distinct = 0;
        for (uint64_t seed = 1; distinct < (1u << log_coef) + 2 && seed < 200; seed++) {
            Proof p = api::generate_proof(d, seed, cfg);
            auto pos = api::verify_samples(p, seed);
            CHECK(pos.has_value());
            if (!pos) break;
            auto ev = p.evaluations();
            for (size_t i = 0; i < pos->size(); i++) {
                positions.push_back((*pos)[i]);  // (repeats across clients stay in the pool: the call ignores them)
                values.push_back(ev[i]);
                if (!have[(*pos)[i]]) have[(*pos)[i]] = true, distinct++;
            }
        }
        CHECK(distinct >= (1u << log_coef) + 2);
        Context& ctx = default_context();
        CHECK(ctx.reconstruct_from_samples(positions, values, log_coef, log_domain, d.size()) == d);
        values[3].v[1] = (values[3].v[1] + 1) % 0x7fffffffu;
        bool refused = false;
        try {
            ctx.reconstruct_from_samples(positions, values, log_coef, log_domain, d.size());
        } catch (const Error&) {
            refused = true;
        }
        CHECK(refused);
    }
    {  // too small a polynomial for the FRI configuration: the reference panics
        std::vector<uint8_t> tiny = {1, 2, 3};
        bool panicked = false;
        try {
            api::generate_proof(tiny, std::nullopt, PCS_CONFIG);
        } catch (const Panic&) {
            panicked = true;
        }
        CHECK(panicked);
    }
    {  // batches: the same results as one call per blob
        const uint32_t count = 5;
        const size_t len = 777, stride = 800;
        std::vector<uint8_t> buf(stride * count, 0xEE);
        for (uint32_t i = 0; i < count; i++)
            for (size_t j = 0; j < len; j++) buf[i * stride + j] = (uint8_t)(31 * i + 7 * j + (j >> 3));
        std::vector<uint64_t> seeds = {11, 12, 13, 14, 15};
        Context& ctx = default_context();
        auto roots = ctx.commit_batch(buf.data(), stride, len, count, 4);
        auto proofs = ctx.commit_and_generate_proof_batch(buf.data(), stride, len, count, seeds.data(), PCS_CONFIG);
        CHECK(roots.size() == count && proofs.size() == count);
        for (uint32_t i = 0; i < count && i < proofs.size(); i++) {
            std::vector<uint8_t> one(buf.begin() + i * stride, buf.begin() + i * stride + len);
            CHECK(roots[i] == api::commit(one, 4));
            auto single = proof::commit_and_generate_proof(one, seeds[i], PCS_CONFIG);
            CHECK(proofs[i].first == single.first && proofs[i].first == roots[i]);
            CHECK(proofs[i].second.serialize() == single.second.serialize());
            CHECK(api::verify(proofs[i].second, seeds[i]));
        }
    }
    {  // host threads: one context per thread (thread_local default_context), proofs produced concurrently, then verified and
       // freed on the main thread — after the producing threads (and their contexts) are gone
        const int n_threads = 4, per_thread = 6;
        std::vector<std::vector<std::pair<Commitment, Proof>>> got(n_threads);
        std::vector<std::thread> ts;
        for (int t = 0; t < n_threads; t++)
            ts.emplace_back([&, t] {
                for (int i = 0; i < per_thread; i++) {
                    std::vector<uint8_t> d(1000 + 97 * t + 13 * i);
                    for (size_t j = 0; j < d.size(); j++) d[j] = (uint8_t)(j * 31 + t * 7 + i);
                    got[t].push_back(proof::commit_and_generate_proof(d, (uint64_t)(100 * t + i), PCS_CONFIG));
                }
            });
        for (auto& th : ts) th.join();
        for (int t = 0; t < n_threads; t++)
            for (int i = 0; i < per_thread; i++) {
                std::vector<uint8_t> d(1000 + 97 * t + 13 * i);
                for (size_t j = 0; j < d.size(); j++) d[j] = (uint8_t)(j * 31 + t * 7 + i);
                CHECK(got[t][i].first == api::commit(d, 4));
                CHECK(api::verify(got[t][i].second, (uint64_t)(100 * t + i)));
                CHECK(!api::verify(got[t][i].second, (uint64_t)(100 * t + i + 1)));
            }
        got.clear();  // frees 24 proofs whose contexts were destroyed with their threads
    }
    {  // options are per context: the general path for a small blob and the host-planned openings give the same bytes as the defaults
        std::vector<uint8_t> d(1024);
        for (size_t j = 0; j < d.size(); j++) d[j] = (uint8_t)(j % 256);  // benches/commit.rs:6-8
        auto want = proof::commit_and_generate_proof(d, (uint64_t)d.size(), PCS_CONFIG);
        Context other(0);
        other.set_option("FRIEDA_NO_SMALL_FUSED", 1);
        other.set_option("FRIEDA_HOST_DECOMMIT", 1);
        auto got = other.commit_and_generate_proof(d.data(), d.size(), (uint64_t)d.size(), PCS_CONFIG);
        CHECK(got.first == want.first && got.second.serialize() == want.second.serialize());
        CHECK(other.commit(d.data(), d.size(), 4) == api::commit(d, 4));
        bool refused = false;
        try {
            other.set_option("FRIEDA_NO_SUCH_OPTION", 1);
        } catch (const Error&) {
            refused = true;
        }
        CHECK(refused);
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}
