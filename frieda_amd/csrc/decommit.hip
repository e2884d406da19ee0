// decommit.hip — query sampling and the openings of a proof, on the device (gfx950).
//
// Replaces, for the prover, `channel.mix_u64(nonce)`, `Queries::generate`, `FriProver::decommit` (per layer
// `compute_decommitment_positions_and_witness_evals` + `MerkleProver::decommit`) and the evaluations gather of
// /root/reference/src/proof.rs:59-66 (stwo core/queries.rs, core/fri.rs, core/vcs/prover.rs).  The host used to do the
// transcript step and the planning between two synchronisations (nonce download -> plan -> gather launch -> download); here
// one launch behind the grind writes the openings in proof order straight into pinned host memory, so a proof needs a single
// synchronisation and no host-side planning.  prover.cpp keeps the host planner as the fallback (more than 1024 queries, an
// opening list that does not fit LDS, FRIEDA_HOST_DECOMMIT=1) and as the cross-check in the tests.
//
// Structure of the openings.  Let uq be the sorted, de-duplicated queries (positions in the 2^n circle domain), U_s =
// unique(uq >> s), and E_s = the children (at shift s - 1) of the nodes of U_s that are NOT in U_{s-1}: a node of U_s always
// has one or two children in U_{s-1}, so it contributes at most one entry, and E_s is ascending.  FRI layer li (log size
// n - li; li = 0 is the circle evaluation) is queried at U_li; because folding a query and walking one tree level up are the
// same shift, every list the reference builds for that layer is one of the E_s:
//   fri_witness   = the values at E_{li+1}                      (the pair members the verifier cannot derive)
//   hash_witness  = for s = li+2 .. n: the hashes of E_s at tree level n - s + 1   (bottom-up, left to right)
// (the leaf level contributes no hashes: both members of every queried pair are opened).  So the kernel computes the E_s
// once — one wave per s, ballot + popcount compaction — and every layer's output is a set of slices of that table.
#include <hip/hip_runtime.h>

#include "dev_transcript.h"
#include "kernels.h"
#include "tree_dev.h"

namespace frieda {
namespace k {

namespace {

constexpr int DC_THREADS = 256;
constexpr uint32_t DC_MAX_Q = DECOMMIT_MAX_QUERIES;  // 1024
constexpr uint32_t DC_E_CAP = 11264;                 // n * (unique queries) slots for the E_s tables (44 KiB of LDS)

__device__ __forceinline__ bool emit_of(const uint32_t* u, uint32_t nu, uint32_t i, uint32_t s, uint32_t& child) {
    if (i >= nu) return false;
    const uint32_t x = u[i], v = x >> s, bit = (x >> (s - 1)) & 1u;
    const bool first = i == 0 || (u[i - 1] >> s) != v;
    const bool last = i + 1 == nu || (u[i + 1] >> s) != v;
    if (first && bit) {  // every position below v lies in the right child: the left one is missing
        child = 2 * v;
        return true;
    }
    if (last && !bit) {  // every position below v lies in the left child
        child = 2 * v + 1;
        return true;
    }
    return false;
}

// A node of the two levels above the leaves, re-hashed from the layer's values (4 columns of 2^m words): the large trees of a proof
// do not store these levels (tree.hip TreeArgs::skip_bc).  level = log2 of the level's size: m - 1 (the parent of leaves 2 c, 2 c + 1)
// or m - 2 (of leaves 4 c .. 4 c + 3): three or seven compressions in the plain form.
__device__ void node_from_values(const uint32_t* __restrict__ v, uint32_t m, uint32_t level, uint32_t child, uint32_t (&h)[8]) {
    const size_t cs = (size_t)1 << m;
    const uint32_t nb = level + 1 == m ? 1u : 2u;  // level-(m-1) nodes under the requested one
    uint32_t hb[2][8];
#pragma unroll
    for (uint32_t q = 0; q < 2; q++) {
        if (q < nb) {
            const size_t b = nb == 1 ? child : 2 * (size_t)child + q, j = 2 * b;
            uint32_t l[8], r[8], mm[16];
            treedev::leaf_hash<B2_LAT>(v[j], v[cs + j], v[2 * cs + j], v[3 * cs + j], l);
            treedev::leaf_hash<B2_LAT>(v[j + 1], v[cs + j + 1], v[2 * cs + j + 1], v[3 * cs + j + 1], r);
#pragma unroll
            for (int w = 0; w < 8; w++) mm[w] = l[w], mm[8 + w] = r[w];
            b2_merkle_block<B2_LAT>(mm, hb[q]);
        }
    }
    if (nb == 1) {
#pragma unroll
        for (int w = 0; w < 8; w++) h[w] = hb[0][w];
    } else {
        uint32_t mm[16];
#pragma unroll
        for (int w = 0; w < 8; w++) mm[w] = hb[0][w], mm[8 + w] = hb[1][w];
        b2_merkle_block<B2_LAT>(mm, h);
    }
}

__global__ __launch_bounds__(DC_THREADS) void decommit_kernel(DecommitArgs a) {
#ifndef FRIEDA_NO_LATENCY_PRIO
    __builtin_amdgcn_s_setprio(3);  // a latency chain that may share the chip with another proof's wide kernels (tree.hip)
#endif
    __shared__ uint32_t s_q[DC_MAX_Q];        // raw draws, then sorted
    __shared__ uint32_t s_u[DC_MAX_Q];        // sorted unique queries
    __shared__ uint32_t s_scan[2][DC_MAX_Q];  // prefix sums of the "first occurrence" flags (more than 64 queries only)
    __shared__ uint32_t s_E[DC_E_CAP];        // E_s at [(s - 1) * nu, (s - 1) * nu + |E_s|)
    __shared__ uint32_t s_cnt[64], s_base[66], s_hoff[64];
    __shared__ uint32_t s_digest[8];
    __shared__ uint32_t s_status, s_nu;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t blob = blockIdx.y;
    const size_t boff = (size_t)blob * a.bstride;
    const DevTranscript* tr = a.tr + blob;
    uint8_t* out = a.out + (size_t)blob * a.out_stride;
    uint32_t* hdr = reinterpret_cast<uint32_t*>(out);
    const uint32_t n = a.n, nq = a.n_queries, nl = a.n_layers;
    const unsigned long long lt_mask = (1ull << lane) - 1;

    // ---- channel.mix_u64(nonce) (src/proof.rs:59) ----
    if (t == 0) {
        // both loads first (independent), then Blake2sChannel::mix_u64: the bare compression keyed by the digest
        const unsigned long long nonce = tr->nonce;
        uint32_t h[8], r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = tr->ch.digest[i];
        uint32_t st = 0;
        if (nonce == ~0ull) {
            st = DECOMMIT_NO_NONCE;
        } else {
            const uint32_t m[16] = {(uint32_t)nonce, (uint32_t)(nonce >> 32), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            b2_compress(h, m, 0, 0, 0, 0, r);
#pragma unroll
            for (int i = 0; i < 8; i++) s_digest[i] = r[i];
        }
        s_status = st;
    }
    __syncthreads();
    if (s_status) {
        if (blockIdx.x == 0 && t == 0) hdr[0] = s_status;
        return;
    }
    // ---- Queries::generate: draw d is blake2s256(digest || d): the draws are independent of each other ----
    const uint32_t mask = (1u << n) - 1;
    uint32_t P = 2;
    while (P < nq) P <<= 1;
    for (uint32_t d = t; d < (nq + 7) / 8; d += DC_THREADS) {
        // Blake2sChannel::draw_random_bytes: standard Blake2s-256 of the 64-byte block digest || LE(d) || 0...: one final block
        uint32_t w[16], r[8], h0[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = s_digest[i], w[8 + i] = 0, h0[i] = b2detail::IV[i];
        w[8] = d;
        h0[0] ^= 0x01010020u;
        b2_compress(h0, w, 64, 0, 0xFFFFFFFFu, 0, r);
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (8 * d + j < nq) s_q[8 * d + j] = r[j] & mask;
    }
    for (uint32_t i = nq + t; i < P; i += DC_THREADS) s_q[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (P <= 64) {
        // the usual case: one wave sorts by rank and de-duplicates with a ballot — no barriers
        if (wave == 0) {
            const uint32_t v = s_q[lane < P ? lane : 0];
            uint32_t rank = 0;
#pragma unroll
            for (int j = 0; j < 64; j++) {  // lanes >= P hold a copy of lane 0's value and are masked out
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v, j);
                rank += ((uint32_t)j < P && (o < v || (o == v && (uint32_t)j < lane))) ? 1u : 0u;
            }
            if (lane < P) s_q[rank] = v;
            const uint32_t x = s_q[lane < P ? lane : 0], prev = s_q[lane > 0 && lane < P ? lane - 1 : 0];
            const bool first = lane < nq && (lane == 0 || x != prev);
            const unsigned long long m = __ballot(first);
            if (first) s_u[__popcll(m & lt_mask)] = x;
            if (lane == 0) s_nu = (uint32_t)__popcll(m);
        }
        __syncthreads();
    } else {
        // bitonic sort, then an inclusive scan of the first-occurrence flags (Hillis-Steele) and compaction
        for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t i = t; i < P; i += DC_THREADS) {
                    const uint32_t ixj = i ^ j;
                    if (ixj > i) {
                        const uint32_t x = s_q[i], y = s_q[ixj];
                        const bool asc = (i & k2) == 0;
                        if ((x > y) == asc) {
                            s_q[i] = y;
                            s_q[ixj] = x;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (uint32_t i = t; i < P; i += DC_THREADS) s_scan[0][i] = (i < nq && (i == 0 || s_q[i] != s_q[i - 1])) ? 1u : 0u;
        __syncthreads();
        uint32_t src = 0;
        for (uint32_t off = 1; off < P; off <<= 1) {
            for (uint32_t i = t; i < P; i += DC_THREADS) s_scan[src ^ 1][i] = s_scan[src][i] + (i >= off ? s_scan[src][i - off] : 0u);
            __syncthreads();
            src ^= 1;
        }
        for (uint32_t i = t; i < nq; i += DC_THREADS)
            if (i == 0 || s_q[i] != s_q[i - 1]) s_u[s_scan[src][i] - 1] = s_q[i];
        if (t == 0) s_nu = s_scan[src][P - 1];
        __syncthreads();
    }
    const uint32_t nu = s_nu;
    if (n * nu > DC_E_CAP) {  // uniform: the E tables would not fit — the host plans this proof
        if (blockIdx.x == 0 && t == 0) hdr[0] = DECOMMIT_OVERFLOW;
        return;
    }

    // ---- the E_s tables: one wave per s, entries compacted with ballot + popcount into the slot of s ----
    for (uint32_t s = 1 + wave; s <= n; s += DC_THREADS / 64) {
        uint32_t run = 0;
        for (uint32_t i0 = 0; i0 < nu; i0 += 64) {
            uint32_t child = 0;
            const bool e = emit_of(s_u, nu, i0 + lane, s, child);
            const unsigned long long m = __ballot(e);
            if (e) s_E[(s - 1) * nu + run + (uint32_t)__popcll(m & lt_mask)] = child;
            run += (uint32_t)__popcll(m);
        }
        if (lane == 0) s_cnt[s] = run;
    }
    __syncthreads();
    // prefix sums by one wave: base[s] = |E_1| + ... + |E_{s-1}|;  hash_witness of layer li = E_{li+2} .. E_n, so its size is
    // total - base[li + 2] and hoff[li] = the sizes of the layers before it
    if (wave == 0) {
        const uint32_t c = (lane >= 1 && lane <= n) ? s_cnt[lane] : 0u;  // lane s holds |E_s|
        uint32_t inc = c;
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += o;
        }
        s_base[lane + 1] = inc;  // base[s + 1] = |E_1| + ... + |E_s|
        if (lane == 0) s_base[0] = 0, s_base[1] = 0;
        const uint32_t total = (uint32_t)__shfl((int)inc, 63);
        // lane li: hashes of layer li = total - base[li + 2] = total - (inclusive sum at lane li + 1)
        const uint32_t inc_next = (uint32_t)__shfl_down((int)inc, 1);
        const uint32_t hl = lane < nl ? total - (lane + 1 <= 63 ? inc_next : total) : 0u;
        uint32_t hinc = hl;
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)hinc, off);
            if (lane >= off) hinc += o;
        }
        s_hoff[lane] = hinc - hl;  // exclusive
        const uint32_t n_hashes = (uint32_t)__shfl((int)hinc, 63);
        const uint32_t wit = (uint32_t)__shfl((int)inc, (int)nl);  // |E_1| + ... + |E_nl|  (nl <= n <= 27)
        const uint32_t n_words = 4 * nu + 4 * wit;
        const uint32_t st = (n_words > a.max_words || n_hashes > a.max_hashes) ? (uint32_t)DECOMMIT_OVERFLOW : (uint32_t)DECOMMIT_OK;
        if (lane == 0) s_status = st;
        if (blockIdx.x == 0) {
            if (lane >= 1 && lane <= n) hdr[3 + lane] = c;
            if (lane == 0) {
                hdr[1] = nu;
                hdr[2] = n_words;
                hdr[3] = n_hashes;
                hdr[0] = st;
            }
        }
    }
    __syncthreads();
    if (s_status) return;

    // ---- outputs, in proof order; the workgroups of a blob (gridDim.x) take interleaved slices ----
    const uint32_t gt = blockIdx.x * DC_THREADS + t, gstride = gridDim.x * DC_THREADS;
    uint32_t* ow = reinterpret_cast<uint32_t*>(out + a.words_off);
    uint4* oh = reinterpret_cast<uint4*>(out + a.hashes_off);
    // Proof.evaluations (src/proof.rs:62-66): the four coordinates at every query
    {
        const uint32_t* v0 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.vals[0]) + boff);
        for (uint32_t e = gt; e < 4 * nu; e += gstride) ow[e] = v0[((size_t)(e & 3) << n) + s_u[e >> 2]];
    }
    // fri_witness of layer li: the values at E_{li+1}.  Dense grid (layer, slot, coordinate); empty slots are skipped.
    for (uint32_t e = gt; e < 4 * nu * nl; e += gstride) {
        const uint32_t li = e / (4 * nu), r = e - li * 4 * nu, k = r >> 2, c = r & 3;
        if (k < s_cnt[li + 1]) {
            const uint32_t* v = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.vals[li]) + boff);
            ow[4 * nu + 4 * (s_base[li + 1] + k) + c] = v[((size_t)c << (n - li)) + s_E[li * nu + k]];
        }
    }
    // hash_witness: entry k of E_s is opened in every layer li <= s - 2, at level n - s + 1 of that layer's tree, and lands at
    // hash position hoff[li] + (base[s] + k - base[li + 2]) of the output.  Dense grid (s, slot, 16-byte half); the layers of
    // an entry are independent loads, issued four at a time.
    if (n >= 2) {
        for (uint32_t e = gt; e < 2 * nu * (n - 1); e += gstride) {
            const uint32_t s = 2 + e / (2 * nu), r = e - (s - 2) * 2 * nu, k = r >> 1, half = r & 1;
            if (k >= s_cnt[s]) continue;
            const uint32_t child = s_E[(s - 1) * nu + k], idx = s_base[s] + k;
            const uint32_t level = n - s + 1;
            const uint32_t li_end = s - 1 < nl ? s - 1 : nl;  // layers 0 .. li_end - 1
            for (uint32_t li0 = 0; li0 < li_end; li0 += 4) {
                uint4 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t li = li0 + j;
                    // (a level the tree does not hold is re-hashed below, by the thread of the entry's first half)
                    if (li < li_end && !(n - li >= a.skip_log && level + 2 >= n - li)) {
                        const uint8_t* tree = a.trees[li] + boff + (((size_t)64 << (n - li)) - ((size_t)64 << level));
                        v[j] = reinterpret_cast<const uint4*>(tree + 32 * (size_t)child)[half];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t li = li0 + j;
                    if (li < li_end && !(n - li >= a.skip_log && level + 2 >= n - li)) oh[2 * (size_t)(s_hoff[li] + (idx - s_base[li + 2])) + half] = v[j];
                }
            }
            // the (at most two) layers in which this entry sits one or two levels above the leaves of a tree that keeps neither
            if (half == 0) {
#pragma unroll 1
                for (uint32_t li = (s >= 3 ? s - 3 : 0); li < li_end; li++) {  // level + 2 >= n - li  <=>  li >= s - 3
                    if (n - li < a.skip_log) continue;
                    uint32_t h[8];
                    node_from_values(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.vals[li]) + boff), n - li, level, child, h);
                    uint4* o = oh + 2 * (size_t)(s_hoff[li] + (idx - s_base[li + 2]));
                    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
                    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
                }
            }
        }
    }
}

}  // namespace

void decommit(const Launch& L, const DecommitArgs& a, uint32_t wgs_per_blob) {
    Scope scope(L, "decommit", 0.0);
    decommit_kernel<<<dim3(wgs_per_blob ? wgs_per_blob : 1, L.batch), DC_THREADS, 0, L.stream>>>(a);
}

}  // namespace k
}  // namespace frieda
