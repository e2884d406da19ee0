// channel.h — Fiat–Shamir transcript of the prove / verify path.
//
// Mirrors stwo-prover@19d12d7 core/channel/blake2s.rs `Blake2sChannel` and
// core/vcs/blake2_merkle.rs `Blake2sMerkleChannel::mix_root`, the channel frieda instantiates at
// /root/reference/src/proof.rs:39-42,59,80-83,92-96.  PARITY UNPINNED: no reference test holds a known
// answer for the transcript; everything that depends on it is kept behind this one class
// so a drift found on a cargo-equipped machine is a one-file fix.
//
// Usable from host code and from a single device thread (the on-device FRI tail keeps the transcript in
// registers between layers instead of a host round trip per layer).
#pragma once
#include <stdint.h>

#include "blake2s.h"
#include "field.h"

namespace frieda {

struct Channel {
    uint32_t digest[8];  // little-endian words of the 32-byte digest
    uint32_t n_challenges;
    uint32_t n_sent;

    FR_HD void init() {
        for (int i = 0; i < 8; i++) digest[i] = 0;
        n_challenges = 0;
        n_sent = 0;
    }
    FR_HD void update_digest(const uint32_t (&d)[8]) {
        for (int i = 0; i < 8; i++) digest[i] = d[i];
        n_challenges += 1;
        n_sent = 0;
    }
    // raw compression keyed by the digest, message = [lo, hi, 0 x 14]
    FR_HD void mix_u64(uint64_t v) {
        uint32_t m[16] = {(uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t h[8], r[8];
        for (int i = 0; i < 8; i++) h[i] = digest[i];
        b2_compress(h, m, 0, 0, 0, 0, r);
        update_digest(r);
    }
    // blake2s256(digest || root)
    FR_HD void mix_root(const uint32_t (&root)[8]) {
        uint32_t w[16], r[8];
        for (int i = 0; i < 8; i++) {
            w[i] = digest[i];
            w[8 + i] = root[i];
        }
        b2s256_words(w, 64, r);
        update_digest(r);
    }
    // blake2s256(digest || LE(n_sent) padded to 32 bytes); n_sent += 1
    FR_HD void draw_random_words(uint32_t (&out)[8]) {
        uint32_t w[16];
        for (int i = 0; i < 8; i++) {
            w[i] = digest[i];
            w[8 + i] = 0;
        }
        w[8] = n_sent;
        n_sent += 1;
        b2s256_words(w, 64, out);
    }
    // draw_base_felts retry rule: all eight words < 2P, then reduce; first four form the QM31
    // `bound`: acceptance bound, 2P in stwo; only the retry-branch test passes anything else (<= 2P)
    FR_HD QM31 draw_felt(uint32_t bound = 2u * P31) {
        for (;;) {
            uint32_t w[8];
            draw_random_words(w);
            bool ok = true;
            for (int i = 0; i < 8; i++) ok = ok && (w[i] < bound);
            if (!ok) continue;
            return {m31_reduce_2p(w[0]), m31_reduce_2p(w[1]), m31_reduce_2p(w[2]), m31_reduce_2p(w[3])};
        }
    }
    // u128::from_le_bytes(digest[0..16]).trailing_zeros()
    FR_HD uint32_t trailing_zeros() const {
        uint32_t tz = 0;
        for (int i = 0; i < 4; i++) {
            uint32_t w = digest[i];
            if (w == 0) {
                tz += 32;
                continue;
            }
            while (!(w & 1u)) {
                w >>= 1;
                tz++;
            }
            return tz;
        }
        return 128;
    }
};

}  // namespace frieda
