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
#if defined(__HIP_DEVICE_COMPILE__)
#include "blake2s_asm.h"
#endif

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

#if defined(__HIP_DEVICE_COMPILE__)
// ---- throughput form for the chip-filling kernels (round 5; profiles/r05_issue_pattern.txt, r05_blake2s_runs.txt,
// r05_blake2s_idle_sweep.txt, r05_issue_overlap.txt, r05_blake2s_prio.txt) ----
// What a gfx950 SIMD sustains on this instruction stream depends on HOW the stream is laid out and on WHICH WAVE the arbiter serves,
// not only on its instruction count:
//  * a fine interleave of fast-class (v_xor / v_add) and slow-class (v_alignbit / v_add3) instructions — what the scheduler emits when
//    left alone (average run 2.1) — costs 3975 SIMD cycles per wave-compression (node shape, 8 waves per SIMD);
//  * the same instructions as RUNS of one class (the four columns / diagonals advance one G step at a time): 3860;
//  * with a few IDLE issue states (s_nop) between the runs: 3290 - 3320 (the first half of the round);
//  * with the wave's PRIORITY raised for its slow runs and dropped for its fast runs (s_setprio at the run boundaries): 2310 - 2430.
//    The pipe overlaps a slow instruction of one wave with a fast instruction of ANOTHER wave (four waves of v_alignbit + four of
//    v_xor on a SIMD take 0.72 of the sum of the two alone, tools/issue_overlap.hip), but the arbiter — priority, then age — does not
//    look for such pairs; with the slow runs prioritised the waves in a slow run always win, the others fill in with their fast runs,
//    and the stream issues at ~2.5 cycles per instruction whatever its class.  The other way round (fast runs raised) costs 4000.
// The runs are pinned by data flow: one volatile asm statement takes the four values a step has just written as read-write
// operands, so the step's instructions lie between two such statements at every level of the compiler (a scheduling barrier is not
// enough: IR passes move pure arithmetic across it).  The statement's text is the s_setprio and / or the s_nop; the compiler adds
// an s_nop 0 of its own in front of the next VALU instruction after any inline asm, so N idle states = "s_nop N-2" (N = 1: empty
// text; N = 0 without a priority: no statement).
// IDLE = 0xABC: idle states at the boundaries slow -> fast (A), fast -> slow (B), slow -> slow (C: rotr 7 -> the next add3);
// bits 12 - 13 / 14 - 15: the priorities of the slow runs (b2_half_round_runs).  SET: -1 = the statement leaves the priority
// alone, else what it sets.
template <int N, int SET = -1>
__device__ __forceinline__ void b2_pin(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    static_assert(N >= 0 && N <= 6, "idle states 0 .. 6");  // (more than 6 never paid: profiles/r05_blake2s_idle_sweep.txt)
    static_assert(SET >= -1 && SET <= 3, "wave priorities 0 .. 3");
#define FR_B2_PIN(TXT) asm volatile(TXT : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define FR_B2_PIN_N(PRE)                                    \
    do {                                                    \
        if constexpr (N <= 1) FR_B2_PIN(PRE);               \
        if constexpr (N == 2) FR_B2_PIN(PRE "s_nop 0");     \
        if constexpr (N == 3) FR_B2_PIN(PRE "s_nop 1");     \
        if constexpr (N == 4) FR_B2_PIN(PRE "s_nop 2");     \
        if constexpr (N == 5) FR_B2_PIN(PRE "s_nop 3");     \
        if constexpr (N == 6) FR_B2_PIN(PRE "s_nop 4");     \
    } while (0)
    if constexpr (SET < 0 && N >= 1) FR_B2_PIN_N("");
    if constexpr (SET == 0) FR_B2_PIN_N("s_setprio 0\n\t");
    if constexpr (SET == 1) FR_B2_PIN_N("s_setprio 1\n\t");
    if constexpr (SET == 2) FR_B2_PIN_N("s_setprio 2\n\t");
    if constexpr (SET == 3) FR_B2_PIN_N("s_setprio 3\n\t");
#undef FR_B2_PIN_N
#undef FR_B2_PIN
}
// one half-round (R: round, H: 0 = columns, 1 = diagonals); the four G functions are written out (no loops: the unroller gives up on
// loops that carry inline asm once a kernel holds a few dozen compressions)
template <int IDLE, int R, int H>
__device__ __forceinline__ void b2_half_round_runs(uint32_t (&v)[16], const uint32_t (&m)[16]) {
    using b2detail::SIGMA;
    constexpr int NA = (IDLE >> 8) & 15, NB = (IDLE >> 4) & 15, NC = IDLE & 15, PRIO = (IDLE >> 12) & 3, PADD = (IDLE >> 14) & 3;
    // what the statements set: LO before a fast run, HI before a run of rotates, HA before a run that holds the v_add3 (bits 14 - 15,
    // 0 = as the rotates).  Rotates above add3 above the fast class measures 3 - 4 % below one raised level (r05_blake2s_prio.txt).
    constexpr int LO = PRIO ? 0 : -1, HI = PRIO ? PRIO : -1, HA = PRIO ? (PADD ? PADD : PRIO) : -1;
    constexpr bool FIRST = (R == 0 && H == 0), LAST = (R == 9 && H == 1);
    constexpr int a0 = 0, a1 = 1, a2 = 2, a3 = 3;
    constexpr int b0 = H ? 5 : 4, b1 = H ? 6 : 5, b2 = H ? 7 : 6, b3 = H ? 4 : 7;
    constexpr int c0 = H ? 10 : 8, c1 = H ? 11 : 9, c2 = H ? 8 : 10, c3 = H ? 9 : 11;
    constexpr int d0 = H ? 15 : 12, d1 = H ? 12 : 13, d2 = H ? 13 : 14, d3 = H ? 14 : 15;
    constexpr int o = 8 * H;
#define FR_B2_4(X) X(0) X(1) X(2) X(3)
#define FR_B2_AX(q) v[a##q] = v[a##q] + v[b##q] + m[SIGMA[R][o + 2 * q]];
#define FR_B2_AY(q) v[a##q] = v[a##q] + v[b##q] + m[SIGMA[R][o + 2 * q + 1]];
#define FR_B2_DX(q) v[d##q] ^= v[a##q];
#define FR_B2_CD(q) v[c##q] += v[d##q];
#define FR_B2_BX(q) v[b##q] ^= v[c##q];
#define FR_B2_RD16(q) v[d##q] = b2detail::rotr(v[d##q], 16);
#define FR_B2_RD8(q) v[d##q] = b2detail::rotr(v[d##q], 8);
#define FR_B2_RB12(q) v[b##q] = b2detail::rotr(v[b##q], 12);
#define FR_B2_RB7(q) v[b##q] = b2detail::rotr(v[b##q], 7);
    if constexpr (FIRST && PRIO != 0) b2_pin<0, HA>(v[a0], v[a1], v[a2], v[a3]);  // the compression opens with a slow run
    FR_B2_4(FR_B2_AX)
    b2_pin<NA, LO>(v[a0], v[a1], v[a2], v[a3]);
    FR_B2_4(FR_B2_DX)
    b2_pin<NB, HI>(v[d0], v[d1], v[d2], v[d3]);
    FR_B2_4(FR_B2_RD16)
    b2_pin<NA, LO>(v[d0], v[d1], v[d2], v[d3]);
    FR_B2_4(FR_B2_CD)
    FR_B2_4(FR_B2_BX)
    b2_pin<NB, HA>(v[b0], v[b1], v[b2], v[b3]);
    FR_B2_4(FR_B2_RB12)
    FR_B2_4(FR_B2_AY)
    b2_pin<NA, LO>(v[a0], v[a1], v[a2], v[a3]);
    FR_B2_4(FR_B2_DX)
    b2_pin<NB, HI>(v[d0], v[d1], v[d2], v[d3]);
    FR_B2_4(FR_B2_RD8)
    b2_pin<NA, LO>(v[d0], v[d1], v[d2], v[d3]);
    FR_B2_4(FR_B2_CD)
    FR_B2_4(FR_B2_BX)
    b2_pin<NB, HI>(v[b0], v[b1], v[b2], v[b3]);
    FR_B2_4(FR_B2_RB7)
    // (the next half-round opens with the add3 run; back to priority 0 at the end of the compression)
    b2_pin<LAST ? 0 : NC, LAST ? LO : (HA != HI ? HA : -1)>(v[b0], v[b1], v[b2], v[b3]);
#undef FR_B2_4
#undef FR_B2_AX
#undef FR_B2_AY
#undef FR_B2_DX
#undef FR_B2_CD
#undef FR_B2_BX
#undef FR_B2_RD16
#undef FR_B2_RD8
#undef FR_B2_RB12
#undef FR_B2_RB7
}
template <int IDLE, int R>
__device__ __forceinline__ void b2_round_runs(uint32_t (&v)[16], const uint32_t (&m)[16]) {
    b2_half_round_runs<IDLE, R, 0>(v, m);
    b2_half_round_runs<IDLE, R, 1>(v, m);
}
// out = F(h, m, t, f) in the run-structured form
template <int IDLE>
__device__ __forceinline__ void b2_compress_runs(const uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t t0, uint32_t t1, uint32_t f0, uint32_t f1,
                                                 uint32_t (&out)[8]) {
    using b2detail::IV;
    uint32_t v[16] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], IV[0], IV[1], IV[2], IV[3], IV[4] ^ t0, IV[5] ^ t1, IV[6] ^ f0, IV[7] ^ f1};
    b2_round_runs<IDLE, 0>(v, m);
    b2_round_runs<IDLE, 1>(v, m);
    b2_round_runs<IDLE, 2>(v, m);
    b2_round_runs<IDLE, 3>(v, m);
    b2_round_runs<IDLE, 4>(v, m);
    b2_round_runs<IDLE, 5>(v, m);
    b2_round_runs<IDLE, 6>(v, m);
    b2_round_runs<IDLE, 7>(v, m);
    b2_round_runs<IDLE, 8>(v, m);
    b2_round_runs<IDLE, 9>(v, m);
    out[0] = h[0] ^ v[0] ^ v[8];
    out[1] = h[1] ^ v[1] ^ v[9];
    out[2] = h[2] ^ v[2] ^ v[10];
    out[3] = h[3] ^ v[3] ^ v[11];
    out[4] = h[4] ^ v[4] ^ v[12];
    out[5] = h[5] ^ v[5] ^ v[13];
    out[6] = h[6] ^ v[6] ^ v[14];
    out[7] = h[7] ^ v[7] ^ v[15];
}
// out = F(0, m, 0, 0) (the Merkle shape); message words that are compile-time zeros (a leaf) fold as in the plain form
template <int IDLE>
__device__ __forceinline__ void b2_merkle_block_runs(const uint32_t (&m)[16], uint32_t (&out)[8]) {
    const uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    b2_compress_runs<IDLE>(z, m, 0, 0, 0, 0, out);
}
#endif
// ---- round 6: the same stream as ONE asm block per message shape (blake2s_asm.h, generated by tools/gen_blake2s_asm.py) ----
// Every pin above is an inline-asm statement, and the compiler puts an s_nop 0 behind each one (180 per compression).  With the
// whole compression as one block — the run order and the priorities of the 0xB000 setting, round 0 folded against the zero state
// and the IV by the generator — there is no statement boundary inside and the order no longer depends on how the compiler treats the
// pins: 2330 -> 2270 SIMD cycles per node-shaped wave-compression, 2225 -> 2130 per leaf (8 waves per SIMD; 2510 -> 2390 / 2430 ->
// 2300 at 4; tools/blake2s_asm.hip, profiles/r06_asm_block.txt).  Measured against it and not kept: one raised level for every slow
// run, the add3 runs at 1, the priority per instruction instead of per run, the leaf's two-operand adds after its add3 in a run
// (all within 1 %), rotr 16 as v_pk_add_u16 op_sel (VOP3P: slower than v_alignbit: 2960 / 2790).
// The block form serves the 0xB000 setting only; any other IDLE value (the A/B builds) and -DFRIEDA_B2_NO_ASM take the pinned form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FRIEDA_B2_NO_ASM)
#define FRIEDA_B2_ASM_BLOCK 1
#endif
// settings of the throughput form, per message shape and kernel family (A/B knobs of the build: tools/build_variant.sh <name>
// -DFRIEDA_B2_IDLE_NODE=0x...).  0xQPABC in bit fields: bits 12 - 13 P = wave priority of the runs of rotates, bits 14 - 15 Q = of
// the runs that hold the v_add3 (0 = P), A / B / C = idle states (above).  SIMD cycles per wave-compression, node / leaf shape,
// 8 waves per SIMD (tools/blake2s_runs.hip; profiles/r05_blake2s_idle_sweep.txt, r05_blake2s_prio.txt):
//   the scheduler's own order 3975 / 3810;  idle states alone (0x603, the first form of this round) 3290 - 3320 / 3130 - 3150;
//   one raised priority for every slow run (0x2000) 2370 - 2430 / 2250 - 2290 (2690 / 2460 at 4 waves, 3200 / 2960 at 2);
//   rotates 3, add3 runs 2 (0xB000, the default) 2310 - 2350 / 2190 - 2260;  the idle states on top of the priorities 2440 / 2350.
// In the product (profiles/r05_prio_product_ab.txt) 0x2000, 0x2603 and 0xB000 are within a per cent of each other.
#ifndef FRIEDA_B2_IDLE_NODE
#define FRIEDA_B2_IDLE_NODE 0xB000
#endif
#ifndef FRIEDA_B2_IDLE_LEAF
#define FRIEDA_B2_IDLE_LEAF 0xB000
#endif
// the fused last transform pass + tree launch (ntt.hip: 4 waves per SIMD, 120 VGPRs) has its own pair of knobs
#ifndef FRIEDA_B2_IDLE_NTT_NODE
#define FRIEDA_B2_IDLE_NTT_NODE 0xB000
#endif
#ifndef FRIEDA_B2_IDLE_NTT_LEAF
#define FRIEDA_B2_IDLE_NTT_LEAF 0xB000
#endif
// the grind (one compression per nonce with a chaining value: fri.hip, tree.hip)
#ifndef FRIEDA_B2_IDLE_GRIND
#define FRIEDA_B2_IDLE_GRIND 0xB000
#endif
// (which form a tree launch takes is decided per launch: tree.hip tp_launch — below ~3 waves per SIMD the runs only cost latency)

// Blake2sMerkleHasher::hash_node for one 16-word block from the zero state: the shape of every node of
// frieda's trees (leaf = 4 column words + 12 zero words; inner node = left || right).
FR_HD void b2_merkle_block_lat(const uint32_t (&m)[16], uint32_t (&out)[8]) {  // plain form: host code and the latency-bound (one-workgroup) kernels
    const uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    b2_compress(z, m, 0, 0, 0, 0, out);
}
// the same for the chip-filling kernels: the throughput form on the device (IDLE: see above), the plain form on the host
// F(h, m, t, f) for chip-filling launches with a chaining value (the grind): throughput form on the device, plain form on the host
template <int IDLE>
FR_HD void b2_compress_tp(const uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t t0, uint32_t t1, uint32_t f0, uint32_t f1, uint32_t (&out)[8]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FRIEDA_B2_NO_RUNS)
    b2_compress_runs<IDLE>(h, m, t0, t1, f0, f1, out);
#else
    b2_compress(h, m, t0, t1, f0, f1, out);
#endif
}
// IDLE = B2_LAT selects the plain form (call sites of the latency-bound kernels: one workgroup, or few waves per SIMD, where an idle
// state is pure delay)
constexpr int B2_LAT = -1;
constexpr int B2_ASM_SETTING = 0xB000;  // the setting the generated asm blocks implement
template <int IDLE = FRIEDA_B2_IDLE_NODE>
FR_HD void b2_merkle_block(const uint32_t (&m)[16], uint32_t (&out)[8]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FRIEDA_B2_NO_RUNS)
    if constexpr (IDLE < 0) {
        b2_merkle_block_lat(m, out);
    } else {
#if defined(FRIEDA_B2_ASM_BLOCK)
        if constexpr (IDLE == B2_ASM_SETTING)
            b2_asm_node(m, out);
        else
#endif
            b2_merkle_block_runs<IDLE>(m, out);
    }
#else
    b2_merkle_block_lat(m, out);
#endif
}
// the leaf shape: 4 column words + 12 zero words (their adds fold: at compile time in the C++ forms, in the generator for the block)
template <int IDLE = FRIEDA_B2_IDLE_LEAF>
FR_HD void b2_merkle_leaf(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t (&out)[8]) {
    const uint32_t m[16] = {v0, v1, v2, v3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#if defined(FRIEDA_B2_ASM_BLOCK) && !defined(FRIEDA_B2_NO_RUNS)
    if constexpr (IDLE == B2_ASM_SETTING)
        b2_asm_leaf(m, out);
    else
#endif
        b2_merkle_block<IDLE>(m, out);
}

// ---- the proof-of-work compression (round 6): Blake2sChannel::mix_u64 with the nonce = F(digest, [nonce_lo, nonce_hi, 0 x 14], 0, 0, 0, 0), of
// which GrindOps::grind (src/proof.rs:58) only needs the trailing zeros.  b2_grind_word0 returns word 0 of the result — enough to decide
// trailing_zeros >= pow_bits for pow_bits <= 32 and to pre-select for more (the caller recomputes a survivor in full: one nonce in 2^pow_bits).
// Of round 0's column step only the first quadruple sees the nonce: b2_grind_prepare runs the other three ONCE for a digest (12 state words, plus
// h0 and h4 passed through); the generated block (blake2s_asm.h b2_asm_grind) starts from there and ends the last half-round where out[0] is
// known: 895 instead of ~955 instructions, as one asm block instead of the pinned C++ form (profiles/r06_grind_asm.txt).
FR_HD void b2_grind_prepare(const uint32_t (&h)[8], uint32_t (&pre)[14]) {
    using b2detail::IV;
    uint32_t v1 = h[1], v5 = h[5], v9 = IV[1], v13 = IV[5], v2 = h[2], v6 = h[6], v10 = IV[2], v14 = IV[6], v3 = h[3], v7 = h[7], v11 = IV[3], v15 = IV[7];
    FR_B2_G(v1, v5, v9, v13, 0u, 0u);  // message words 2 .. 7 are zero
    FR_B2_G(v2, v6, v10, v14, 0u, 0u);
    FR_B2_G(v3, v7, v11, v15, 0u, 0u);
    pre[0] = h[0], pre[1] = h[4];
    pre[2] = v1, pre[3] = v5, pre[4] = v9, pre[5] = v13;
    pre[6] = v2, pre[7] = v6, pre[8] = v10, pre[9] = v14;
    pre[10] = v3, pre[11] = v7, pre[12] = v11, pre[13] = v15;
}
// word 0 of F(h, [m0, m1, 0 x 14], 0, 0, 0, 0); `pre` from b2_grind_prepare(h)
FR_HD uint32_t b2_grind_word0(const uint32_t (&h)[8], const uint32_t (&pre)[14], uint32_t m0, uint32_t m1) {
#if defined(FRIEDA_B2_ASM_BLOCK) && !defined(FRIEDA_B2_NO_RUNS) && !defined(FRIEDA_B2_GRIND_FULL)  // (-DFRIEDA_B2_GRIND_FULL: the whole pinned compression, A/B)
    (void)h;
    return b2_asm_grind(m0, m1, pre);
#else
    (void)pre;
    const uint32_t m[16] = {m0, m1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t r[8];
    b2_compress_tp<FRIEDA_B2_IDLE_GRIND>(h, m, 0, 0, 0, 0, r);
    return r[0];
#endif
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
