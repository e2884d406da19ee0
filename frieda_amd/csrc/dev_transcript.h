// dev_transcript.h — Fiat–Shamir state kept in device memory for the whole FRI commit phase.
//
// The reference's FriProver::commit (/root/reference/src/proof.rs:52-57; stwo core/fri.rs) alternates
// "Merkle root -> channel.mix_root -> channel.draw_felt -> fold" once per layer.  Evaluating the channel on the
// host costs a stream synchronise + 32-byte D2H per layer (~20 per proof).  Here the kernel that finishes a tree
// performs the two channel steps in one lane and leaves alpha in this block; the next fold kernel reads it from
// there, so the whole commit phase is one uninterrupted stream of launches.
#pragma once
#include <stdint.h>

#include "channel.h"

namespace frieda {

constexpr uint32_t DT_MAX_LAYERS = 40;
constexpr uint32_t DT_MAX_LAST_POLY = 2048;  // QM31 coefficients of the last-layer polynomial the device path supports

struct DevTranscript {
    Channel ch;             // current channel state
    uint32_t alpha[4];      // folding alpha drawn after the latest root
    uint32_t status;        // 0 ok; bit 0: last layer failed stwo's degree assertion ("invalid degree")
    uint32_t n_roots;       // roots mixed so far
    unsigned long long nonce;  // grind result (atomicMin), ~0 = not found yet
    uint32_t roots[DT_MAX_LAYERS][8];
    uint32_t alphas[DT_MAX_LAYERS][4];
    uint32_t n_last_poly;   // QM31 count of last_poly
    uint32_t draw_bound;    // acceptance bound of draw_felt (2P; a test hook may lower it)
    uint32_t pad_[2];
    uint32_t last_poly[4 * DT_MAX_LAST_POLY];  // LinePoly coefficients, stwo internal order
};

}  // namespace frieda
