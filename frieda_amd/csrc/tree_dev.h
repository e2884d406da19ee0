// tree_dev.h — device helpers shared by the kernels that build Merkle levels (tree.hip, the fused last transform pass in
// ntt.hip): hash storage in global memory (32-byte array of structs, leaves-first level offsets) and in LDS (struct of arrays).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blake2s.h"

namespace frieda {
namespace k {
namespace treedev {

// ---- LDS hash levels, struct-of-arrays: word w of hash j at reg[w * stride + j], stride = count + 4 (even) ----
__device__ __forceinline__ void lds_put(uint32_t* reg, uint32_t stride, uint32_t j, const uint32_t (&h)[8]) {
#pragma unroll
    for (int w = 0; w < 8; w++) reg[w * stride + j] = h[w];
}
__device__ __forceinline__ void lds_children(const uint32_t* reg, uint32_t stride, uint32_t j, uint32_t (&m)[16]) {
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint2 v = *reinterpret_cast<const uint2*>(reg + w * stride + 2 * j);
        m[w] = v.x;
        m[8 + w] = v.y;
    }
}
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
// leaf of 4 column words: the twelve zero message words are compile-time constants, so their adds fold away
template <int IDLE = FRIEDA_B2_IDLE_LEAF>
__device__ __forceinline__ void leaf_hash(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t (&h)[8]) {
    b2_merkle_leaf<IDLE>(v0, v1, v2, v3, h);
}
__device__ __forceinline__ size_t layer_off(uint32_t tree_log, uint32_t layer) {
    return ((size_t)64 << tree_log) - ((size_t)64 << layer);
}

}  // namespace treedev
}  // namespace k
}  // namespace frieda
