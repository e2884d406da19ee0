// transcript.cpp — the variable-length parts of the Fiat–Shamir transcript shared by prover and verifier:
// Blake2sChannel::mix_felts and stwo core/queries.rs::Queries::{generate, fold}.  PARITY UNPINNED (see channel.h).
#include <algorithm>

#include "host.h"

namespace frieda {

// Blake2sChannel::mix_felts: blake2s256(digest || LE words of every QM31)
void channel_mix_felts(Channel& ch, const std::vector<QM31>& felts) {
    std::vector<uint32_t> w(8 + 4 * felts.size());
    for (int i = 0; i < 8; i++) w[i] = ch.digest[i];
    for (size_t i = 0; i < felts.size(); i++) {
        w[8 + 4 * i] = felts[i].a;
        w[9 + 4 * i] = felts[i].b;
        w[10 + 4 * i] = felts[i].c;
        w[11 + 4 * i] = felts[i].d;
    }
    uint32_t len = (uint32_t)(4 * w.size());
    w.resize((w.size() + 15) / 16 * 16, 0u);
    uint32_t r[8];
    b2s256_words(w.data(), len, r);
    ch.update_digest(r);
}

// Queries::generate
std::vector<uint32_t> generate_queries(Channel& ch, uint32_t log_domain_size, uint32_t n_queries) {
    std::vector<uint32_t> q;
    uint32_t mask = (1u << log_domain_size) - 1;
    while (q.size() < n_queries) {
        uint32_t w[8];
        ch.draw_random_words(w);
        for (int i = 0; i < 8 && q.size() < n_queries; i++) q.push_back(w[i] & mask);
    }
    std::sort(q.begin(), q.end());
    q.erase(std::unique(q.begin(), q.end()), q.end());
    return q;
}
// Queries::fold
std::vector<uint32_t> fold_queries(const std::vector<uint32_t>& q, uint32_t n_folds) {
    std::vector<uint32_t> r;
    for (uint32_t v : q)
        if (r.empty() || r.back() != (v >> n_folds)) r.push_back(v >> n_folds);
    return r;
}

}  // namespace frieda
