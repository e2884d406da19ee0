// proof.cpp — canonical little-endian wire image of frieda::proof::Proof.
//
// The reference derives serde on Proof (/root/reference/src/proof.rs:19) but depends on no serializer
// crate, so it defines no wire format; this layout is ours.  Field order follows the Rust
// struct nesting: config, log_size_bound, proof_of_work, evaluations, first layer, inner layers, last
// layer polynomial.
#include <string.h>

#include "host.h"

namespace frieda {

namespace {

constexpr uint32_t MAGIC = 0x41445246u;  // "FRDA"
constexpr uint32_t VERSION = 1;

struct Writer {
    std::vector<uint8_t> b;
    void u32(uint32_t v) {
        for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void qm(const QM31& q) {
        u32(q.a);
        u32(q.b);
        u32(q.c);
        u32(q.d);
    }
    void bytes(const uint8_t* p, size_t n) { b.insert(b.end(), p, p + n); }
    void layer(const LayerProof& l) {
        bytes(l.commitment.data(), 32);
        u32((uint32_t)l.fri_witness.size());
        for (auto& q : l.fri_witness) qm(q);
        u32((uint32_t)l.hash_witness.size());
        for (auto& h : l.hash_witness) bytes(h.data(), 32);
        u32((uint32_t)l.column_witness.size());
        for (auto v : l.column_witness) u32(v);
    }
};

struct Reader {
    const uint8_t* p;
    size_t len, pos = 0;
    bool ok = true;
    uint32_t u32() {
        if (pos + 4 > len) {
            ok = false;
            return 0;
        }
        uint32_t v = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
        pos += 4;
        return v;
    }
    // a count that must be backed by `unit` bytes per element of remaining input
    size_t count(size_t unit) {
        uint32_t c = u32();
        if (!ok || (size_t)c * unit > len - pos) {
            ok = false;
            return 0;
        }
        return c;
    }
    QM31 qm() {
        QM31 q;
        q.a = u32();
        q.b = u32();
        q.c = u32();
        q.d = u32();
        return q;
    }
    void bytes(uint8_t* dst, size_t n) {
        if (pos + n > len) {
            ok = false;
            return;
        }
        memcpy(dst, p + pos, n);
        pos += n;
    }
    void layer(LayerProof& l) {
        bytes(l.commitment.data(), 32);
        size_t nf = count(16);
        l.fri_witness.resize(nf);
        for (auto& q : l.fri_witness) q = qm();
        size_t nh = count(32);
        l.hash_witness.resize(nh);
        for (auto& h : l.hash_witness) bytes(h.data(), 32);
        size_t nc = count(4);
        l.column_witness.resize(nc);
        for (auto& v : l.column_witness) v = u32();
    }
};

}  // namespace

std::vector<uint8_t> serialize_proof(const ProofData& p) {
    Writer w;
    w.u32(MAGIC);
    w.u32(VERSION);
    w.u32(p.pcs_config.pow_bits);
    w.u32(p.pcs_config.log_blowup_factor);
    w.u32(p.pcs_config.log_last_layer_degree_bound);
    w.u32(p.pcs_config.n_queries);
    w.u32(p.log_size_bound);
    w.u32((uint32_t)p.proof_of_work);
    w.u32((uint32_t)(p.proof_of_work >> 32));
    w.u32((uint32_t)p.evaluations.size());
    for (auto& q : p.evaluations) w.qm(q);
    w.layer(p.first_layer);
    w.u32((uint32_t)p.inner_layers.size());
    for (auto& l : p.inner_layers) w.layer(l);
    w.u32((uint32_t)p.last_layer_poly.size());
    for (auto& q : p.last_layer_poly) w.qm(q);
    return std::move(w.b);
}

bool deserialize_proof(const uint8_t* buf, size_t len, ProofData& out) {
    Reader r{buf, len};
    if (r.u32() != MAGIC || r.u32() != VERSION) return false;
    out.pcs_config.pow_bits = r.u32();
    out.pcs_config.log_blowup_factor = r.u32();
    out.pcs_config.log_last_layer_degree_bound = r.u32();
    out.pcs_config.n_queries = r.u32();
    out.log_size_bound = r.u32();
    uint64_t lo = r.u32(), hi = r.u32();
    out.proof_of_work = lo | (hi << 32);
    size_t ne = r.count(16);
    out.evaluations.resize(ne);
    for (auto& q : out.evaluations) q = r.qm();
    r.layer(out.first_layer);
    size_t ni = r.count(44);  // a layer is at least commitment + three counts
    out.inner_layers.resize(ni);
    for (auto& l : out.inner_layers) {
        r.layer(l);
        if (!r.ok) return false;
    }
    size_t nl = r.count(16);
    out.last_layer_poly.resize(nl);
    for (auto& q : out.last_layer_poly) q = r.qm();
    return r.ok && r.pos == len;
}

}  // namespace frieda
