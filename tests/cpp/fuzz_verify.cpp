// fuzz_verify.cpp — host-only robustness check of the proof parser and verifier, built with AddressSanitizer +
// UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool).  Reads a valid wire image, then
// for `iters` rounds applies random bit flips / truncations / splices and runs frieda_proof_deserialize + frieda_verify.
// Any memory error aborts; accepted mutants are counted (a mutant that still verifies must be byte-identical in meaning,
// e.g. a flip in an ignored position — there are none in this format, so the count must be 0).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "frieda_hip.h"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const bool has_seed = argv[2][0] != '-';
    uint64_t seed = has_seed ? std::strtoull(argv[2], nullptr, 10) : 0;
    const int iters = std::atoi(argv[3]);
    frieda_proof* p = nullptr;
    int ok = 0;
    if (frieda_proof_deserialize(img.data(), img.size(), &p) != FRIEDA_OK) return 3;
    if (frieda_verify(p, has_seed ? &seed : nullptr, &ok) != FRIEDA_OK || !ok) return 4;
    frieda_proof_free(p);
    std::mt19937_64 rng(12345);
    int parsed = 0, accepted = 0, panics = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> m = img;
        switch (rng() % 4) {
            case 0: {  // bit flips
                int k = 1 + rng() % 4;
                for (int i = 0; i < k; i++) m[rng() % m.size()] ^= (uint8_t)(1u << (rng() % 8));
                break;
            }
            case 1: m.resize(rng() % m.size()); break;                   // truncate
            case 2: {                                                       // overwrite a count-like word with junk
                size_t pos = (rng() % (m.size() / 4)) * 4;
                uint32_t v = (uint32_t)rng();
                for (int b = 0; b < 4; b++) m[pos + b] = (uint8_t)(v >> (8 * b));
                break;
            }
            default: {  // splice a chunk elsewhere
                size_t a = rng() % m.size(), b = rng() % m.size(), n = rng() % 64;
                for (size_t i = 0; i < n && a + i < m.size() && b + i < m.size(); i++) m[a + i] = m[b + i];
            }
        }
        frieda_proof* q = nullptr;
        if (frieda_proof_deserialize(m.data(), m.size(), &q) != FRIEDA_OK) continue;
        parsed++;
        int good = 0;
        int rc = frieda_verify(q, has_seed ? &seed : nullptr, &good);
        if (rc == FRIEDA_ERR_INVARIANT) panics++;
        if (rc == FRIEDA_OK && good && m != img) accepted++;
        frieda_proof_free(q);
    }
    std::printf("iters %d parsed %d panics %d accepted_mutants %d\n", iters, parsed, panics, accepted);
    return accepted ? 5 : 0;
}
