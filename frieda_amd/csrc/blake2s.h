// blake2s.h — Blake2s compression function (RFC 7693 F) for host code and gfx950 kernels.
//
// Two users, two different constructions:
//  * Merkle nodes: stwo-prover@19d12d7 core/vcs/blake2_merkle.rs `Blake2sMerkleHasher::hash_node` —
//    the *bare* compression function chained from an all-zero state with t = f = 0, no parameter block,
//    no length, no finalisation flag (pinned by the golden root of /root/reference/src/commit.rs:31-37).
//  * Fiat–Shamir channel: standard Blake2s-256 (blake2 0.10.6) except Blake2sChannel::mix_u64, which is
//    again the bare compression keyed by the current digest.
//
// The rounds are fully unrolled with the message permutation resolved at compile time so that the
// sixteen state words and sixteen message words live in VGPRs (no scratch, no LDS): per G function
// 4 adds (two of them v_add3_u32), 4 xors and 4 v_alignbit_b32 rotates.
#pragma once
#include <stdint.h>

#include "field.h"

namespace frieda {

struct B2State {
    uint32_t h[8];
};

namespace b2detail {
constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                            0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
constexpr uint8_t SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

FR_HD uint32_t rotr(uint32_t x, uint32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_rotateright32(x, r);
#else
    return (x >> r) | (x << (32 - r));
#endif
}
}  // namespace b2detail

#define FR_B2_G(a, b, c, d, x, y)      \
    do {                               \
        a = a + b + (x);               \
        d = b2detail::rotr(d ^ a, 16); \
        c = c + d;                     \
        b = b2detail::rotr(b ^ c, 12); \
        a = a + b + (y);               \
        d = b2detail::rotr(d ^ a, 8);  \
        c = c + d;                     \
        b = b2detail::rotr(b ^ c, 7);  \
    } while (0)

template <int R>
FR_HD void b2_round(uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3, uint32_t& v4, uint32_t& v5, uint32_t& v6,
                    uint32_t& v7, uint32_t& v8, uint32_t& v9, uint32_t& v10, uint32_t& v11, uint32_t& v12, uint32_t& v13,
                    uint32_t& v14, uint32_t& v15, const uint32_t (&m)[16]) {
    using b2detail::SIGMA;
    FR_B2_G(v0, v4, v8, v12, m[SIGMA[R][0]], m[SIGMA[R][1]]);
    FR_B2_G(v1, v5, v9, v13, m[SIGMA[R][2]], m[SIGMA[R][3]]);
    FR_B2_G(v2, v6, v10, v14, m[SIGMA[R][4]], m[SIGMA[R][5]]);
    FR_B2_G(v3, v7, v11, v15, m[SIGMA[R][6]], m[SIGMA[R][7]]);
    FR_B2_G(v0, v5, v10, v15, m[SIGMA[R][8]], m[SIGMA[R][9]]);
    FR_B2_G(v1, v6, v11, v12, m[SIGMA[R][10]], m[SIGMA[R][11]]);
    FR_B2_G(v2, v7, v8, v13, m[SIGMA[R][12]], m[SIGMA[R][13]]);
    FR_B2_G(v3, v4, v9, v14, m[SIGMA[R][14]], m[SIGMA[R][15]]);
}

// out = F(h, m, t, f)
FR_HD void b2_compress(const uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t t0, uint32_t t1, uint32_t f0,
                       uint32_t f1, uint32_t (&out)[8]) {
    using b2detail::IV;
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = IV[0], v9 = IV[1], v10 = IV[2], v11 = IV[3];
    uint32_t v12 = IV[4] ^ t0, v13 = IV[5] ^ t1, v14 = IV[6] ^ f0, v15 = IV[7] ^ f1;
    b2_round<0>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<1>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<2>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<3>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<4>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<5>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<6>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<7>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<8>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    b2_round<9>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m);
    out[0] = h[0] ^ v0 ^ v8;
    out[1] = h[1] ^ v1 ^ v9;
    out[2] = h[2] ^ v2 ^ v10;
    out[3] = h[3] ^ v3 ^ v11;
    out[4] = h[4] ^ v4 ^ v12;
    out[5] = h[5] ^ v5 ^ v13;
    out[6] = h[6] ^ v6 ^ v14;
    out[7] = h[7] ^ v7 ^ v15;
}

// Blake2sMerkleHasher::hash_node for one 16-word block from the zero state: the shape of every node of
// frieda's trees (leaf = 4 column words + 12 zero words; inner node = left || right).
FR_HD void b2_merkle_block(const uint32_t (&m)[16], uint32_t (&out)[8]) {
    const uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    b2_compress(z, m, 0, 0, 0, 0, out);
}

// Standard unkeyed Blake2s-256 of a message given as little-endian words, at most one... any number of
// 64-byte blocks; `len` in bytes, `words` must be zero padded to a multiple of 16 words.
FR_HD void b2s256_words(const uint32_t* words, uint32_t len, uint32_t (&out)[8]) {
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = b2detail::IV[i];
    h[0] ^= 0x01010020u;
    uint32_t off = 0;
    while (len - off > 64) {
        uint32_t m[16], nx[8];
        for (int i = 0; i < 16; i++) m[i] = words[off / 4 + i];
        b2_compress(h, m, off + 64, 0, 0, 0, nx);
        for (int i = 0; i < 8; i++) h[i] = nx[i];
        off += 64;
    }
    uint32_t m[16];
    for (int i = 0; i < 16; i++) m[i] = (off + 4 * i < len) ? words[off / 4 + i] : 0u;
    b2_compress(h, m, len, 0, 0xFFFFFFFFu, 0, out);
}

}  // namespace frieda
