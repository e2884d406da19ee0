// verifier.cpp — host-side FRI verifier: api::verify (/root/reference/src/lib.rs:41, src/proof.rs:79-101).
//
// Follows stwo-prover@19d12d7 core/fri.rs (FriVerifier::{commit, sample_query_positions, decommit},
// FriFirstLayerVerifier / FriInnerLayerVerifier::verify_and_fold, SparseEvaluation::{fold_circle, fold_line}),
// core/vcs/verifier.rs (MerkleVerifier::verify), core/queries.rs and core/poly/line.rs
// (LinePoly::eval_at_point).  PARITY UNPINNED beyond the reference's own accept / reject tests
// (src/proof.rs:136-193).  The reference verifier is O(n_queries * log N) hashes — microseconds — so it stays on the
// host by design; there is no device code on this path.
#include <algorithm>
#include <string.h>

#include "host.h"

namespace frieda {

Hash32 hash_node_host(const uint8_t* left, const uint8_t* right, const uint32_t* values, size_t n_values) {
    // Blake2sMerkleHasher::hash_node
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nx[8], m[16];
    auto rd = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    if (left) {
        for (int i = 0; i < 8; i++) {
            m[i] = rd(left + 4 * i);
            m[8 + i] = rd(right + 4 * i);
        }
        b2_compress(st, m, 0, 0, 0, 0, nx);
        memcpy(st, nx, sizeof st);
    }
    for (size_t off = 0; off < n_values; off += 16) {
        for (size_t i = 0; i < 16; i++) m[i] = off + i < n_values ? values[off + i] : 0u;
        b2_compress(st, m, 0, 0, 0, 0, nx);
        memcpy(st, nx, sizeof st);
    }
    Hash32 out;
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(st[i] >> (8 * b));
    return out;
}

namespace {

void words_of(const Hash32& h, uint32_t (&w)[8]) {
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)h[4 * i] | ((uint32_t)h[4 * i + 1] << 8) | ((uint32_t)h[4 * i + 2] << 16) | ((uint32_t)h[4 * i + 3] << 24);
}

struct SparseEvaluation {
    std::vector<std::array<QM31, 2>> subset_evals;
    std::vector<uint32_t> subset_start;  // position of the subset's first member in the (bit-reversed) layer
    std::vector<uint32_t> decommitment_positions;
};

enum class Rebuild { Ok, InsufficientWitness, QueryEvalsExhausted };

// compute_decommitment_positions_and_rebuild_evals with fold_step = 1
Rebuild rebuild_evals(const std::vector<uint32_t>& queries, const std::vector<QM31>& query_evals,
                      const std::vector<QM31>& witness, size_t& witness_used, SparseEvaluation& se) {
    size_t qi = 0, wi = 0;
    for (size_t i = 0; i < queries.size();) {
        size_t j = i;
        while (j < queries.size() && (queries[j] >> 1) == (queries[i] >> 1)) j++;
        uint32_t start = (queries[i] >> 1) << 1;
        std::array<QM31, 2> ev{};
        size_t kq = i;
        for (uint32_t pos = start; pos < start + 2; pos++) {
            se.decommitment_positions.push_back(pos);
            if (kq < j && queries[kq] == pos) {
                kq++;
                if (qi >= query_evals.size()) return Rebuild::QueryEvalsExhausted;  // `.next().unwrap()` upstream
                ev[pos - start] = query_evals[qi++];
            } else {
                if (wi >= witness.size()) return Rebuild::InsufficientWitness;
                ev[pos - start] = witness[wi++];
            }
        }
        se.subset_evals.push_back(ev);
        se.subset_start.push_back(start);
        i = j;
    }
    witness_used = wi;
    return Rebuild::Ok;
}

// MerkleVerifier::verify for a tree whose 4 columns sit on the leaf layer
bool merkle_verify(const Hash32& root, uint32_t log_size, const std::vector<uint32_t>& positions,
                   const std::vector<uint32_t>& queried_values, const LayerProof& lp) {
    size_t hw = 0, vi = 0;
    std::vector<std::pair<uint32_t, Hash32>> last;
    for (int layer = (int)log_size; layer >= 0; layer--) {
        bool leaf = layer == (int)log_size;
        std::vector<std::pair<uint32_t, Hash32>> cur;
        size_t pi = 0, ci = 0;
        const size_t n_col = leaf ? positions.size() : 0;
        while (pi < last.size() || ci < n_col) {
            uint32_t node;
            if (pi < last.size() && ci < n_col)
                node = std::min(last[pi].first / 2, positions[ci]);
            else if (pi < last.size())
                node = last[pi].first / 2;
            else
                node = positions[ci];
            const uint8_t *lh = nullptr, *rh = nullptr;
            if (!leaf) {
                if (pi < last.size() && last[pi].first == 2 * node)
                    lh = last[pi++].second.data();
                else if (hw < lp.hash_witness.size())
                    lh = lp.hash_witness[hw++].data();
                else
                    return false;  // WitnessTooShort
                if (pi < last.size() && last[pi].first == 2 * node + 1)
                    rh = last[pi++].second.data();
                else if (hw < lp.hash_witness.size())
                    rh = lp.hash_witness[hw++].data();
                else
                    return false;
            }
            uint32_t vals[4];
            size_t nv = 0;
            if (leaf) {
                ci++;  // every leaf node of this shape is a queried column position
                if (vi + 4 > queried_values.size()) return false;  // TooFewQueriedValues
                for (; nv < 4; nv++) vals[nv] = queried_values[vi++];
            }
            cur.emplace_back(node, hash_node_host(lh, rh, vals, nv));
        }
        last.swap(cur);
    }
    if (hw != lp.hash_witness.size()) return false;  // WitnessTooLong
    if (vi != queried_values.size()) return false;   // TooManyQueriedValues
    if (!lp.column_witness.empty()) return false;    // WitnessTooLong
    return last.size() == 1 && last[0].second == root;
}

std::vector<uint32_t> flatten(const SparseEvaluation& se) {
    std::vector<uint32_t> v;
    for (auto& s : se.subset_evals)
        for (auto& q : s) {
            v.push_back(q.a);
            v.push_back(q.b);
            v.push_back(q.c);
            v.push_back(q.d);
        }
    return v;
}

// LinePoly::eval_at_point; coeffs in stwo's internal (bit-reversed) order, length 2^log
QM31 line_poly_eval(const QM31* coeffs, size_t n, const QM31* factors) {
    if (n == 1) return coeffs[0];
    QM31 l = line_poly_eval(coeffs, n / 2, factors + 1), r = line_poly_eval(coeffs + n / 2, n / 2, factors + 1);
    return qm_add(l, qm_mul(r, factors[0]));
}

}  // namespace

int verify(const ProofData& proof, const uint64_t* seed, int* ok, std::vector<uint32_t>* out_queries) {
    *ok = 0;
    if (out_queries) out_queries->clear();
    const frieda_pcs_config& cfg = proof.pcs_config;
    const uint32_t B = cfg.log_blowup_factor, last = cfg.log_last_layer_degree_bound, L = proof.log_size_bound;
    // CirclePolyDegreeBound::fold_to_line underflows (panics) for L == 0; domain sizes outside the group do too
    if (L < 1 || L + B < 2 || L + B > 30) return FRIEDA_ERR_INVARIANT;
    const uint32_t n = L + B;

    Channel ch;
    ch.init();
    if (seed) ch.mix_u64(*seed);  // src/proof.rs:81-83

    // FriVerifier::commit
    uint32_t w[8];
    words_of(proof.first_layer.commitment, w);
    ch.mix_root(w);
    std::vector<QM31> alphas{ch.draw_felt()};
    uint32_t bound = L - 1;
    for (auto& lp : proof.inner_layers) {
        words_of(lp.commitment, w);
        ch.mix_root(w);
        alphas.push_back(ch.draw_felt());
        if (bound < 1) return FRIEDA_OK;  // InvalidNumFriLayers
        bound -= 1;
    }
    if (bound != last) return FRIEDA_OK;                                          // InvalidNumFriLayers
    if (proof.last_layer_poly.size() > ((size_t)1 << last)) return FRIEDA_OK;     // LastLayerDegreeInvalid
    channel_mix_felts(ch, proof.last_layer_poly);

    // src/proof.rs:92-95
    ch.mix_u64(proof.proof_of_work);
    if (ch.trailing_zeros() < cfg.pow_bits) return FRIEDA_OK;

    // sample_query_positions
    std::vector<uint32_t> queries = generate_queries(ch, n, cfg.n_queries);

    // decommit_first_layer
    SparseEvaluation se;
    size_t used = 0;
    switch (rebuild_evals(queries, proof.evaluations, proof.first_layer.fri_witness, used, se)) {
        case Rebuild::QueryEvalsExhausted: return FRIEDA_ERR_INVARIANT;  // src/proof.rs:166-173 should_panic
        case Rebuild::InsufficientWitness: return FRIEDA_OK;             // FirstLayerEvaluationsInvalid
        case Rebuild::Ok: break;
    }
    if (used != proof.first_layer.fri_witness.size()) return FRIEDA_OK;
    if (!merkle_verify(proof.first_layer.commitment, n, se.decommitment_positions, flatten(se), proof.first_layer)) return FRIEDA_OK;

    // decommit_inner_layers
    if (proof.inner_layers.empty()) return FRIEDA_ERR_INVARIANT;  // assert!(first_layer_columns.is_empty()) upstream
    std::vector<uint32_t> lq = fold_queries(queries, 1);
    std::vector<QM31> evals(lq.size());
    {
        // SparseEvaluation::fold_circle + accumulate_line onto zeros
        Coset h = Coset::half_odds(n - 1);
        for (size_t s = 0; s < se.subset_start.size(); s++) {
            uint32_t pos = bit_reverse(se.subset_start[s], n);  // < N/2: the half coset itself
            CPoint p = h.at(pos);
            QM31 a = se.subset_evals[s][0], b = se.subset_evals[s][1];
            QM31 f0 = qm_add(a, b), f1 = qm_scale(qm_sub(a, b), m31_inv(p.y));
            evals[s] = qm_add(qm_mul(alphas[0], f1), f0);
        }
    }
    uint32_t m = n - 1;
    for (size_t kx = 0; kx < proof.inner_layers.size(); kx++) {
        const LayerProof& lp = proof.inner_layers[kx];
        SparseEvaluation s2;
        size_t u2 = 0;
        if (rebuild_evals(lq, evals, lp.fri_witness, u2, s2) != Rebuild::Ok) return FRIEDA_OK;  // InnerLayerEvaluationsInvalid
        if (u2 != lp.fri_witness.size()) return FRIEDA_OK;
        if (!merkle_verify(lp.commitment, m, s2.decommitment_positions, flatten(s2), lp)) return FRIEDA_OK;
        Coset c = line_coset(n, m);
        std::vector<QM31> folded(s2.subset_start.size());
        for (size_t s = 0; s < s2.subset_start.size(); s++) {
            uint32_t x = c.at(bit_reverse(s2.subset_start[s], m)).x;
            QM31 a = s2.subset_evals[s][0], b = s2.subset_evals[s][1];
            QM31 f0 = qm_add(a, b), f1 = qm_scale(qm_sub(a, b), m31_inv(x));
            folded[s] = qm_add(f0, qm_mul(alphas[kx + 1], f1));
        }
        lq = fold_queries(lq, 1);
        evals.swap(folded);
        m--;
    }

    // decommit_last_layer
    size_t np = proof.last_layer_poly.size();
    if (np == 0 || (np & (np - 1))) return FRIEDA_OK;
    uint32_t plog = 0;
    while (((size_t)1 << plog) < np) plog++;
    Coset c = line_coset(n, m);
    for (size_t i = 0; i < lq.size(); i++) {
        uint32_t x = c.at(bit_reverse(lq[i], m)).x;
        QM31 fac[32];
        QM31 xx{x, 0, 0, 0};
        for (uint32_t d = 0; d < plog; d++) {
            fac[d] = xx;
            QM31 sq = qm_mul(xx, xx);
            xx = qm_sub(qm_add(sq, sq), QM31{1, 0, 0, 0});
        }
        if (!qm_eq(evals[i], line_poly_eval(proof.last_layer_poly.data(), np, fac))) return FRIEDA_OK;  // LastLayerEvaluationsInvalid
    }
    *ok = 1;
    if (out_queries) *out_queries = queries;
    return FRIEDA_OK;
}

}  // namespace frieda
