// merkle.hip — layered Blake2s-compression Merkle trees (gfx950).
//
// Replaces `MerkleProver::<CpuBackend, Blake2sMerkleHasher>::commit` / `CpuBackend::commit_on_layer` /
// `Blake2sMerkleHasher::hash_node` (/root/reference/src/commit.rs:17-21, inside FriProver at
// src/proof.rs:52; stwo core/vcs/{prover,blake2_merkle,blake2s_ref}.rs).  Node i of a layer is
// hash_node((prev[2i], prev[2i+1])?, [col[i] for col in columns]) with the bare compression function
// chained from a zero state (see blake2s.h).  frieda's trees carry their 4 columns on the leaf layer only.
//
// Hash storage is array-of-structs: node i of a layer is the 32 bytes at 32*i, so the two children of a
// parent are one contiguous 64-byte block and a whole wave reads a contiguous 4 KiB span.
// Per node the work is one compression (~970 integer VALU ops) against 64 B read + 32 B written, which puts
// the single-layer kernels at the crossover of the VALU and HBM ceilings.  These are the trait-granular kernels behind
// `frieda_merkle_commit_layer`; whole trees go through the fused multi-level kernels of tree.hip.
#include <hip/hip_runtime.h>

#include "blake2s.h"
#include "kernels.h"

namespace frieda {
namespace k {

namespace {

constexpr int MK_THREADS = 256;

__device__ __forceinline__ void store_hash(uint8_t* out, size_t i, const uint32_t (&h)[8]) {
    uint4* o = reinterpret_cast<uint4*>(out + 32 * i);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

__device__ __forceinline__ void load_children(const uint8_t* prev, size_t i, uint32_t (&m)[16]) {
    const uint4* p = reinterpret_cast<const uint4*>(prev + 64 * i);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    m[0] = a.x, m[1] = a.y, m[2] = a.z, m[3] = a.w;
    m[4] = b.x, m[5] = b.y, m[6] = b.z, m[7] = b.w;
    m[8] = c.x, m[9] = c.y, m[10] = c.z, m[11] = c.w;
    m[12] = d.x, m[13] = d.y, m[14] = d.z, m[15] = d.w;
}

__global__ __launch_bounds__(MK_THREADS) void merkle_leaf4_kernel(const uint32_t* __restrict__ c0,
                                                                  const uint32_t* __restrict__ c1,
                                                                  const uint32_t* __restrict__ c2,
                                                                  const uint32_t* __restrict__ c3, size_t n,
                                                                  uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t h[8];
    b2_merkle_leaf(c0[i], c1[i], c2[i], c3[i], h);
    store_hash(out, i, h);
}

__global__ __launch_bounds__(MK_THREADS) void merkle_node_kernel(const uint8_t* __restrict__ prev, size_t n,
                                                                 uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t m[16], h[8];
    load_children(prev, i, m);
    b2_merkle_block(m, h);
    store_hash(out, i, h);
}

// general commit_on_layer: optional children, then the column values in 16-word blocks
__global__ __launch_bounds__(MK_THREADS) void merkle_generic_kernel(const uint8_t* __restrict__ prev,
                                                                    const uint32_t* const* __restrict__ cols,
                                                                    uint32_t ncols, size_t n, uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * MK_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nx[8], m[16];
    if (prev) {
        load_children(prev, i, m);
        b2_compress(st, m, 0, 0, 0, 0, nx);
        for (int q = 0; q < 8; q++) st[q] = nx[q];
    }
    for (uint32_t off = 0; off < ncols; off += 16) {
#pragma unroll
        for (uint32_t q = 0; q < 16; q++) m[q] = (off + q < ncols) ? cols[off + q][i] : 0u;
        b2_compress(st, m, 0, 0, 0, 0, nx);
        for (int q = 0; q < 8; q++) st[q] = nx[q];
    }
    store_hash(out, i, st);
}


}  // namespace

void merkle_leaf4(const Launch& L, const uint32_t* c0, const uint32_t* c1, const uint32_t* c2, const uint32_t* c3, size_t n,
                  uint8_t* d_out) {
    hipStream_t s = L.stream;
    Scope scope(L, "merkle_leaf4", 48.0 * (double)n);  // 16 B of columns in, 32 B hash out per leaf
    merkle_leaf4_kernel<<<(unsigned)((n + MK_THREADS - 1) / MK_THREADS), MK_THREADS, 0, s>>>(c0, c1, c2, c3, n, d_out);
}

void merkle_node(const Launch& L, const uint8_t* d_prev, size_t n, uint8_t* d_out) {
    hipStream_t s = L.stream;
    Scope scope(L, "merkle_node", 96.0 * (double)n);  // 64 B of children in, 32 B hash out per node
    merkle_node_kernel<<<(unsigned)((n + MK_THREADS - 1) / MK_THREADS), MK_THREADS, 0, s>>>(d_prev, n, d_out);
}

void merkle_layer_generic(const Launch& L, const uint8_t* d_prev, const uint32_t* const* d_col_ptrs, uint32_t ncols, size_t n,
                          uint8_t* d_out) {
    hipStream_t s = L.stream;
    Scope scope(L, "merkle_generic", (double)n * (32.0 + (d_prev ? 64.0 : 0.0) + 4.0 * ncols));
    merkle_generic_kernel<<<(unsigned)((n + MK_THREADS - 1) / MK_THREADS), MK_THREADS, 0, s>>>(d_prev, d_col_ptrs, ncols, n,
                                                                                              d_out);
}

}  // namespace k
}  // namespace frieda
