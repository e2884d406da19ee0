// bench_api.cpp — the reference's criterion benches (/root/reference/benches/commit.rs:4-17, benches/proof.rs:14-61) over include/frieda.hpp:
// groups `commit`, `generate_proof`, `commit_and_generate_proof`, `verify_proof`, the same five inputs ((i % 256) as u8 for 1024 / 4096 / 16384 /
// 65536 bytes + the `blob` fixture), the same configuration (PCS_CONFIG of benches/proof.rs:5-12, seed = Some(data.len())), one call at a time
// from host memory — what `cargo bench` times on the CPU path, here through the C ABI on the GPU.  Not criterion's statistics: a warm-up
// and the mean over a fixed wall-time budget per input, printed one line per (group, input) like criterion's ids.
// Built by __graft_entry__.build(); argv[1] = path of the `blob` fixture, argv[2] (optional) = seconds per measurement (default 0.3).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "frieda.hpp"

using namespace frieda;
using clk = std::chrono::steady_clock;

static const PcsConfig PCS_CONFIG{20, FriConfig{4, 0, 20}};  // benches/proof.rs:5-12

template <typename F>
static double time_ns(F f, double budget_s) {
    f();  // warm-up (workspace, twiddles)
    f();
    size_t iters = 0;
    const auto t0 = clk::now();
    double el = 0;
    do {
        f();
        iters++;
        el = std::chrono::duration<double>(clk::now() - t0).count();
    } while (el < budget_s);
    return 1e9 * el / (double)iters;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const double budget = argc > 2 ? atof(argv[2]) : 0.3;
    std::vector<std::vector<uint8_t>> datas;
    for (size_t size : {1024u, 4096u, 16384u, 65536u}) {
        std::vector<uint8_t> d(size);
        for (size_t i = 0; i < size; i++) d[i] = (uint8_t)(i % 256);
        datas.push_back(d);
    }
    {
        std::ifstream f(argv[1], std::ios::binary);
        datas.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (datas.back().size() != 262146) {
            std::fprintf(stderr, "blob fixture not found at %s\n", argv[1]);
            return 2;
        }
    }
    int bad = 0;
    for (const auto& data : datas) {
        const uint64_t seed = data.size();
        const double t_commit = time_ns([&] { (void)api::commit(data, 4); }, budget);
        const double t_gen = time_ns([&] { (void)api::generate_proof(data, seed, PCS_CONFIG); }, budget);
        const double t_cgp = time_ns([&] { (void)proof::commit_and_generate_proof(data, seed, PCS_CONFIG); }, budget);
        auto [commitment, proof] = proof::commit_and_generate_proof(data, seed, PCS_CONFIG);
        bool ok = true;
        const double t_verify = time_ns([&] { ok = ok && api::verify(proof, seed); }, budget);  // (verify_proof takes a clone upstream: host only)
        if (!ok || commitment != api::commit(data, 4)) bad++;
        std::printf("commit/%zu                      time: %10.1f us\n", data.size(), t_commit / 1e3);
        std::printf("generate_proof/%zu              time: %10.1f us\n", data.size(), t_gen / 1e3);
        std::printf("commit_and_generate_proof/%zu   time: %10.1f us\n", data.size(), t_cgp / 1e3);
        std::printf("verify_proof/%zu                time: %10.1f us\n", data.size(), t_verify / 1e3);
    }
    std::printf("%s\n", bad ? "FAILED: a proof did not verify or a commitment differed" : "ok: every proof verified, every first-layer commitment == commit()");
    return bad ? 1 : 0;
}
