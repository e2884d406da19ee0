// blake2s_runs.hip — round 5: does the RUN STRUCTURE of the instruction stream change the Blake2s compression rate?
// tools/issue_pattern2.hip: a fine interleave of fast-class (v_xor / v_add) and slow-class (v_alignbit / v_add3) instructions costs
// 4.1 - 4.3 SIMD cycles per wave-instruction, runs of 16 same-class instructions 3.9, runs of 64 3.75.  One compression has runs
// of at most 4 (its four columns / diagonals); K independent compressions per thread, interleaved step by step, give runs of 4 K
// to 8 K.  Here: the real compression (frieda_amd/csrc/blake2s.h order as the compiler schedules it = baseline) against K = 1, 2, 4
// compressions advanced in lock step with a scheduling barrier (MODE 2) or a data-flow pin (MODE 3) between the steps (so the runs
// survive the compiler),
// node-shaped (16 message words) and leaf-shaped (4 words + 12 zeros).  Every lane chains compressions on register-resident data.
// Build: hipcc -O3 --offload-arch=gfx950 -Ifrieda_amd/csrc tools/blake2s_runs.hip -o tools/blake2s_runs.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blake2s.h"

using namespace frieda;

struct Stamp {
    unsigned long long c0, r0, c1, r1;
};
__device__ __forceinline__ void stamp_pair(unsigned long long& c, unsigned long long& r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
}

#define SB() __builtin_amdgcn_sched_barrier(0)

// BAR == 2: the runs are pinned by DATA FLOW instead: one empty volatile asm takes the 4 K values a step has just written as
// read-write operands, so the step's instructions lie between two such statements at every level of the compiler (IR passes move
// pure arithmetic across a scheduling barrier — the leaf shape, whose constant message words fold, lost its runs that way)
// NOP: what the pin statement itself contains — 0: nothing (the compiler still puts an s_nop 0 in front of the next VALU instruction, its
// conservative hazard rule after inline asm), 1: "s_nop 0", 2: "s_nop 1", 3: "s_nop 3", 4: nothing AND pins only two values (so two
// statements per step: a nop inside every run of four as well)
#define PIN_ASM4(txt, a, b, c, d) asm volatile(txt : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
template <int K, int NOP>
__device__ __forceinline__ void pin_step(uint32_t (&v)[16][K], const int (&ix)[4]) {
    static_assert(K == 1 || NOP == 0, "nop variants are K = 1 only");
    if constexpr (K == 1) {
        if constexpr (NOP == 0) PIN_ASM4("", v[ix[0]][0], v[ix[1]][0], v[ix[2]][0], v[ix[3]][0]);
        if constexpr (NOP == 1) PIN_ASM4("s_nop 0", v[ix[0]][0], v[ix[1]][0], v[ix[2]][0], v[ix[3]][0]);
        if constexpr (NOP == 2) PIN_ASM4("s_nop 1", v[ix[0]][0], v[ix[1]][0], v[ix[2]][0], v[ix[3]][0]);
        if constexpr (NOP == 3) PIN_ASM4("s_nop 3", v[ix[0]][0], v[ix[1]][0], v[ix[2]][0], v[ix[3]][0]);
        if constexpr (NOP == 4) {
            asm volatile("" : "+v"(v[ix[0]][0]), "+v"(v[ix[1]][0]));
            asm volatile("" : "+v"(v[ix[2]][0]), "+v"(v[ix[3]][0]));
        }
    } else if constexpr (K == 2) {
        asm volatile("" : "+v"(v[ix[0]][0]), "+v"(v[ix[1]][0]), "+v"(v[ix[2]][0]), "+v"(v[ix[3]][0]), "+v"(v[ix[0]][1]), "+v"(v[ix[1]][1]), "+v"(v[ix[2]][1]),
                     "+v"(v[ix[3]][1]));
    } else {
        for (int k = 0; k < K; k += 2)
            asm volatile("" : "+v"(v[ix[0]][k]), "+v"(v[ix[1]][k]), "+v"(v[ix[2]][k]), "+v"(v[ix[3]][k]), "+v"(v[ix[0]][k + 1]), "+v"(v[ix[1]][k + 1]),
                         "+v"(v[ix[2]][k + 1]), "+v"(v[ix[3]][k + 1]));
    }
}

// one G step over the four (a, b, c, d) index quadruples of a half-round, for K compressions at once, as runs of same-class
// instructions; x[q], y[q]: message word indices of quadruple q (compile-time)
template <int K, int LEAF, int BAR>
__device__ __forceinline__ void half_round(uint32_t (&v)[16][K], const uint32_t (&m)[16][K], const int (&ia)[4], const int (&ib)[4], const int (&ic)[4],
                                           const int (&id)[4], const uint8_t* sx) {
#define PIN(ix)                      \
    do {                             \
        if (BAR == 1) SB();          \
        if constexpr (BAR >= 2) pin_step<K, BAR - 2>(v, ix); \
    } while (0)
#define FORQK for (int q = 0; q < 4; q++) for (int k = 0; k < K; k++)
    _Pragma("unroll") FORQK {
        const int x = sx[2 * q];
        v[ia[q]][k] = v[ia[q]][k] + v[ib[q]][k] + ((LEAF && x >= 4) ? 0u : m[x][k]);
    }
    PIN(ia);
    _Pragma("unroll") FORQK v[id[q]][k] ^= v[ia[q]][k];
    PIN(id);
    _Pragma("unroll") FORQK v[id[q]][k] = b2detail::rotr(v[id[q]][k], 16);
    PIN(id);
    _Pragma("unroll") FORQK v[ic[q]][k] += v[id[q]][k];
    _Pragma("unroll") FORQK v[ib[q]][k] ^= v[ic[q]][k];
    PIN(ib);
    _Pragma("unroll") FORQK v[ib[q]][k] = b2detail::rotr(v[ib[q]][k], 12);
    _Pragma("unroll") FORQK {
        const int y = sx[2 * q + 1];
        v[ia[q]][k] = v[ia[q]][k] + v[ib[q]][k] + ((LEAF && y >= 4) ? 0u : m[y][k]);
    }
    PIN(ia);
    _Pragma("unroll") FORQK v[id[q]][k] ^= v[ia[q]][k];
    PIN(id);
    _Pragma("unroll") FORQK v[id[q]][k] = b2detail::rotr(v[id[q]][k], 8);
    PIN(id);
    _Pragma("unroll") FORQK v[ic[q]][k] += v[id[q]][k];
    _Pragma("unroll") FORQK v[ib[q]][k] ^= v[ic[q]][k];
    PIN(ib);
    _Pragma("unroll") FORQK v[ib[q]][k] = b2detail::rotr(v[ib[q]][k], 7);
    if constexpr (BAR >= 2) pin_step<K, BAR - 2>(v, ib);
#undef FORQK
#undef PIN
}

template <int K, int LEAF, int BAR>
__device__ __forceinline__ void compress_k(const uint32_t (&m)[16][K], uint32_t (&out)[8][K]) {
    using b2detail::IV;
    using b2detail::SIGMA;
    uint32_t v[16][K];
    for (int k = 0; k < K; k++) {
        for (int i = 0; i < 8; i++) v[i][k] = 0u;
        for (int i = 0; i < 8; i++) v[8 + i][k] = IV[i];
    }
    constexpr int ca[4] = {0, 1, 2, 3}, cb[4] = {4, 5, 6, 7}, cc[4] = {8, 9, 10, 11}, cd[4] = {12, 13, 14, 15};
    constexpr int db[4] = {5, 6, 7, 4}, dc[4] = {10, 11, 8, 9}, dd[4] = {15, 12, 13, 14};
#pragma unroll
    for (int r = 0; r < 10; r++) {
        half_round<K, LEAF, BAR>(v, m, ca, cb, cc, cd, &SIGMA[r][0]);
        half_round<K, LEAF, BAR>(v, m, ca, db, dc, dd, &SIGMA[r][8]);
    }
    for (int k = 0; k < K; k++)
        for (int i = 0; i < 8; i++) out[i][k] = v[i][k] ^ v[8 + i][k];
}

// MODE 0: the product's b2_merkle_block, K compressions one after another (compiler order); MODE 1: lock step, no barriers (the
// compiler may still interleave as it likes); MODE 2: lock step with scheduling barriers between the runs
template <int K, int LEAF, int MODE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void chain_kernel(uint32_t* out, int iters, Stamp* st) {
    uint32_t m[16][K], h[8][K];
    for (int k = 0; k < K; k++)
        for (int i = 0; i < 16; i++) m[i][k] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x + 977u * k;
    for (int k = 0; k < K; k++)
        for (int i = 0; i < 8; i++) h[i][k] = 0;
    unsigned long long c0, r0, c1, r1;
    stamp_pair(c0, r0);
    asm volatile("" : "+v"(m[0][0]) : "s"(c0));
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                uint32_t mm[16], hh[8];
                for (int i = 0; i < 16; i++) mm[i] = (LEAF && i >= 4) ? 0u : m[i][k];
                b2_merkle_block(mm, hh);
                for (int i = 0; i < 8; i++) h[i][k] = hh[i];
            }
        } else {
            compress_k<K, LEAF, MODE == 2 ? 1 : (MODE >= 3 ? MODE - 1 : 0)>(m, h);
        }
        for (int k = 0; k < K; k++) {
            if (LEAF) {
                for (int i = 0; i < 4; i++) m[i][k] = h[i][k] ^ h[4 + i][k];
            } else {
                for (int i = 0; i < 8; i++) {
                    m[i][k] ^= h[i][k];
                    m[8 + i][k] += h[i][k];
                }
            }
        }
    }
    asm volatile("" ::"v"(h[0][0]), "v"(h[7][K - 1]));
    stamp_pair(c1, r1);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, r0, c1, r1};
    uint32_t s = 0;
    for (int k = 0; k < K; k++)
        for (int i = 0; i < 8; i++) s += h[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- K = 1, per-boundary idle states: N wait states (0 = no pin at all, the two runs merge; n >= 1: a data-flow pin whose text is
// "s_nop n-2" for n >= 2 — the compiler itself puts an s_nop 0 in front of the next VALU instruction after any inline asm) at the
// boundaries slow -> fast (NA), fast -> slow (NB) and slow -> slow (NC: rotr 7 -> the next half-round's add3) ----
template <int N>
__device__ __forceinline__ void pin_n(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    if constexpr (N == 1) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 2) asm volatile("s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 3) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 4) asm volatile("s_nop 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 5) asm volatile("s_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 6) asm volatile("s_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 7) asm volatile("s_nop 5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 8) asm volatile("s_nop 6" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 9) asm volatile("s_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 10) asm volatile("s_nop 8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 12) asm volatile("s_nop 10" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
template <int LEAF, int NA, int NB, int NC>
__device__ __forceinline__ void half_round_n(uint32_t (&v)[16], const uint32_t (&m)[16], const int (&ia)[4], const int (&ib)[4], const int (&ic)[4],
                                             const int (&id)[4], const uint8_t* sx) {
#define P(N, ix) pin_n<N>(v[ix[0]], v[ix[1]], v[ix[2]], v[ix[3]])
#define FQ for (int q = 0; q < 4; q++)
    _Pragma("unroll") FQ {
        const int x = sx[2 * q];
        v[ia[q]] = v[ia[q]] + v[ib[q]] + ((LEAF && x >= 4) ? 0u : m[x]);
    }
    P(NA, ia);
    _Pragma("unroll") FQ v[id[q]] ^= v[ia[q]];
    P(NB, id);
    _Pragma("unroll") FQ v[id[q]] = b2detail::rotr(v[id[q]], 16);
    P(NA, id);
    _Pragma("unroll") FQ v[ic[q]] += v[id[q]];
    _Pragma("unroll") FQ v[ib[q]] ^= v[ic[q]];
    P(NB, ib);
    _Pragma("unroll") FQ v[ib[q]] = b2detail::rotr(v[ib[q]], 12);
    _Pragma("unroll") FQ {
        const int y = sx[2 * q + 1];
        v[ia[q]] = v[ia[q]] + v[ib[q]] + ((LEAF && y >= 4) ? 0u : m[y]);
    }
    P(NA, ia);
    _Pragma("unroll") FQ v[id[q]] ^= v[ia[q]];
    P(NB, id);
    _Pragma("unroll") FQ v[id[q]] = b2detail::rotr(v[id[q]], 8);
    P(NA, id);
    _Pragma("unroll") FQ v[ic[q]] += v[id[q]];
    _Pragma("unroll") FQ v[ib[q]] ^= v[ic[q]];
    P(NB, ib);
    _Pragma("unroll") FQ v[ib[q]] = b2detail::rotr(v[ib[q]], 7);
    P(NC, ib);
#undef FQ
#undef P
}
template <int LEAF, int WAVES, int NA, int NB, int NC>
__global__ __launch_bounds__(256, WAVES) void chain_n_kernel(uint32_t* out, int iters, Stamp* st) {
    using b2detail::IV;
    using b2detail::SIGMA;
    uint32_t m[16], h[8];
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int i = 0; i < 8; i++) h[i] = 0;
    unsigned long long c0, r0, c1, r1;
    stamp_pair(c0, r0);
    asm volatile("" : "+v"(m[0]) : "s"(c0));
    constexpr int ca[4] = {0, 1, 2, 3}, cb[4] = {4, 5, 6, 7}, cc[4] = {8, 9, 10, 11}, cd[4] = {12, 13, 14, 15};
    constexpr int db[4] = {5, 6, 7, 4}, dc[4] = {10, 11, 8, 9}, dd[4] = {15, 12, 13, 14};
    for (int it = 0; it < iters; it++) {
        uint32_t v[16];
        for (int i = 0; i < 8; i++) v[i] = 0u, v[8 + i] = IV[i];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            half_round_n<LEAF, NA, NB, NC>(v, m, ca, cb, cc, cd, &SIGMA[r][0]);
            half_round_n<LEAF, NA, NB, NC>(v, m, ca, db, dc, dd, &SIGMA[r][8]);
        }
        for (int i = 0; i < 8; i++) h[i] = v[i] ^ v[8 + i];
        if (LEAF) {
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    asm volatile("" ::"v"(h[0]), "v"(h[7]));
    stamp_pair(c1, r1);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, r0, c1, r1};
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- the same with TWO compressions in lock step (runs of 8 - 16), idle states per boundary kind as above ----
template <int N>
__device__ __forceinline__ void pin_n8(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t& e, uint32_t& f, uint32_t& g, uint32_t& h) {
    if constexpr (N == 1) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if constexpr (N == 2) asm volatile("s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if constexpr (N == 3) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if constexpr (N == 4) asm volatile("s_nop 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if constexpr (N == 5) asm volatile("s_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if constexpr (N == 6) asm volatile("s_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
template <int LEAF, int NA, int NB, int NC>
__device__ __forceinline__ void half_round_n2(uint32_t (&v)[16][2], const uint32_t (&m)[16][2], const int (&ia)[4], const int (&ib)[4], const int (&ic)[4],
                                              const int (&id)[4], const uint8_t* sx) {
#define P(N, ix) pin_n8<N>(v[ix[0]][0], v[ix[1]][0], v[ix[2]][0], v[ix[3]][0], v[ix[0]][1], v[ix[1]][1], v[ix[2]][1], v[ix[3]][1])
#define FQ for (int q = 0; q < 4; q++) for (int k = 0; k < 2; k++)
    _Pragma("unroll") FQ {
        const int x = sx[2 * q];
        v[ia[q]][k] = v[ia[q]][k] + v[ib[q]][k] + ((LEAF && x >= 4) ? 0u : m[x][k]);
    }
    P(NA, ia);
    _Pragma("unroll") FQ v[id[q]][k] ^= v[ia[q]][k];
    P(NB, id);
    _Pragma("unroll") FQ v[id[q]][k] = b2detail::rotr(v[id[q]][k], 16);
    P(NA, id);
    _Pragma("unroll") FQ v[ic[q]][k] += v[id[q]][k];
    _Pragma("unroll") FQ v[ib[q]][k] ^= v[ic[q]][k];
    P(NB, ib);
    _Pragma("unroll") FQ v[ib[q]][k] = b2detail::rotr(v[ib[q]][k], 12);
    _Pragma("unroll") FQ {
        const int y = sx[2 * q + 1];
        v[ia[q]][k] = v[ia[q]][k] + v[ib[q]][k] + ((LEAF && y >= 4) ? 0u : m[y][k]);
    }
    P(NA, ia);
    _Pragma("unroll") FQ v[id[q]][k] ^= v[ia[q]][k];
    P(NB, id);
    _Pragma("unroll") FQ v[id[q]][k] = b2detail::rotr(v[id[q]][k], 8);
    P(NA, id);
    _Pragma("unroll") FQ v[ic[q]][k] += v[id[q]][k];
    _Pragma("unroll") FQ v[ib[q]][k] ^= v[ic[q]][k];
    P(NB, ib);
    _Pragma("unroll") FQ v[ib[q]][k] = b2detail::rotr(v[ib[q]][k], 7);
    P(NC, ib);
#undef FQ
#undef P
}
template <int LEAF, int WAVES, int NA, int NB, int NC>
__global__ __launch_bounds__(256, WAVES) void chain_n2_kernel(uint32_t* out, int iters, Stamp* st) {
    using b2detail::IV;
    using b2detail::SIGMA;
    uint32_t m[16][2], h[8][2];
    for (int k = 0; k < 2; k++) {
        for (int i = 0; i < 16; i++) m[i][k] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x + 977u * k;
        for (int i = 0; i < 8; i++) h[i][k] = 0;
    }
    unsigned long long c0, r0, c1, r1;
    stamp_pair(c0, r0);
    asm volatile("" : "+v"(m[0][0]) : "s"(c0));
    constexpr int ca[4] = {0, 1, 2, 3}, cb[4] = {4, 5, 6, 7}, cc[4] = {8, 9, 10, 11}, cd[4] = {12, 13, 14, 15};
    constexpr int db[4] = {5, 6, 7, 4}, dc[4] = {10, 11, 8, 9}, dd[4] = {15, 12, 13, 14};
    for (int it = 0; it < iters; it++) {
        uint32_t v[16][2];
        for (int k = 0; k < 2; k++)
            for (int i = 0; i < 8; i++) v[i][k] = 0u, v[8 + i][k] = IV[i];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            half_round_n2<LEAF, NA, NB, NC>(v, m, ca, cb, cc, cd, &SIGMA[r][0]);
            half_round_n2<LEAF, NA, NB, NC>(v, m, ca, db, dc, dd, &SIGMA[r][8]);
        }
        for (int k = 0; k < 2; k++) {
            for (int i = 0; i < 8; i++) h[i][k] = v[i][k] ^ v[8 + i][k];
            if (LEAF) {
                for (int i = 0; i < 4; i++) m[i][k] = h[i][k] ^ h[4 + i][k];
            } else {
                for (int i = 0; i < 8; i++) {
                    m[i][k] ^= h[i][k];
                    m[8 + i][k] += h[i][k];
                }
            }
        }
    }
    asm volatile("" ::"v"(h[0][0]), "v"(h[7][1]));
    stamp_pair(c1, r1);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, r0, c1, r1};
    uint32_t s = 0;
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 8; i++) s += h[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- K = 1, idle states per boundary POSITION of the half-round (nine of them, in order: after a+=b+x, after d^=a, after rotr16, after
// c+=d / b^=c, after rotr12 / a+=b+y, after d^=a, after rotr8, after c+=d / b^=c, after rotr7) ----
template <int LEAF, int... P>
__device__ __forceinline__ void half_round_p(uint32_t (&v)[16], const uint32_t (&m)[16], const int (&ia)[4], const int (&ib)[4], const int (&ic)[4],
                                             const int (&id)[4], const uint8_t* sx) {
    constexpr int N[9] = {P...};
#define PP(i, ix) pin_n<N[i]>(v[ix[0]], v[ix[1]], v[ix[2]], v[ix[3]])
#define FQ for (int q = 0; q < 4; q++)
    _Pragma("unroll") FQ {
        const int x = sx[2 * q];
        v[ia[q]] = v[ia[q]] + v[ib[q]] + ((LEAF && x >= 4) ? 0u : m[x]);
    }
    PP(0, ia);
    _Pragma("unroll") FQ v[id[q]] ^= v[ia[q]];
    PP(1, id);
    _Pragma("unroll") FQ v[id[q]] = b2detail::rotr(v[id[q]], 16);
    PP(2, id);
    _Pragma("unroll") FQ v[ic[q]] += v[id[q]];
    _Pragma("unroll") FQ v[ib[q]] ^= v[ic[q]];
    PP(3, ib);
    _Pragma("unroll") FQ v[ib[q]] = b2detail::rotr(v[ib[q]], 12);
    _Pragma("unroll") FQ {
        const int y = sx[2 * q + 1];
        v[ia[q]] = v[ia[q]] + v[ib[q]] + ((LEAF && y >= 4) ? 0u : m[y]);
    }
    PP(4, ia);
    _Pragma("unroll") FQ v[id[q]] ^= v[ia[q]];
    PP(5, id);
    _Pragma("unroll") FQ v[id[q]] = b2detail::rotr(v[id[q]], 8);
    PP(6, id);
    _Pragma("unroll") FQ v[ic[q]] += v[id[q]];
    _Pragma("unroll") FQ v[ib[q]] ^= v[ic[q]];
    PP(7, ib);
    _Pragma("unroll") FQ v[ib[q]] = b2detail::rotr(v[ib[q]], 7);
    PP(8, ib);
#undef FQ
#undef PP
}
template <int LEAF, int WAVES, int... P>
__global__ __launch_bounds__(256, WAVES) void chain_p_kernel(uint32_t* out, int iters, Stamp* st) {
    using b2detail::IV;
    using b2detail::SIGMA;
    uint32_t m[16], h[8];
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int i = 0; i < 8; i++) h[i] = 0;
    unsigned long long c0, r0, c1, r1;
    stamp_pair(c0, r0);
    asm volatile("" : "+v"(m[0]) : "s"(c0));
    constexpr int ca[4] = {0, 1, 2, 3}, cb[4] = {4, 5, 6, 7}, cc[4] = {8, 9, 10, 11}, cd[4] = {12, 13, 14, 15};
    constexpr int db[4] = {5, 6, 7, 4}, dc[4] = {10, 11, 8, 9}, dd[4] = {15, 12, 13, 14};
    for (int it = 0; it < iters; it++) {
        uint32_t v[16];
        for (int i = 0; i < 8; i++) v[i] = 0u, v[8 + i] = IV[i];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            half_round_p<LEAF, P...>(v, m, ca, cb, cc, cd, &SIGMA[r][0]);
            half_round_p<LEAF, P...>(v, m, ca, db, dc, dd, &SIGMA[r][8]);
        }
        for (int i = 0; i < 8; i++) h[i] = v[i] ^ v[8 + i];
        if (LEAF) {
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    asm volatile("" ::"v"(h[0]), "v"(h[7]));
    stamp_pair(c1, r1);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, r0, c1, r1};
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- K = 1, wave PRIORITY switched at the run boundaries (s_setprio), with or without the idle states of 6-0-3 ----
// V: 1 = slow runs at priority 2, fast runs at 0, with idle states; 2 = the opposite, with idle states; 3 / 4 = the same two without
template <int V, int KIND>  // KIND: 0 = after a slow run that a fast run follows, 1 = after a fast run, 2 = slow -> slow
__device__ __forceinline__ void pin_prio(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    if constexpr (V == 1 && KIND == 0) asm volatile("s_setprio 0\n\ts_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 1 && KIND == 1) asm volatile("s_setprio 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 1 && KIND == 2) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 2 && KIND == 0) asm volatile("s_setprio 2\n\ts_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 2 && KIND == 1) asm volatile("s_setprio 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 2 && KIND == 2) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 3 && KIND == 0) asm volatile("s_setprio 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 3 && KIND == 1) asm volatile("s_setprio 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 3 && KIND == 2) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 4 && KIND == 0) asm volatile("s_setprio 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 4 && KIND == 1) asm volatile("s_setprio 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (V == 4 && KIND == 2) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    // KIND 3 = before a run of rotates only (KIND 1 then = before the rotate-12 + add3 run, KIND 2 = before the add3 run of the next half-round)
    // V 5: rotate runs at 3, runs with add3 at 2, fast 0;  6: rotate runs 2, add3 runs 3;  7: slow 2, fast 1
    if constexpr (V <= 4 && KIND == 3) pin_prio<V, 1>(a, b, c, d);
    // KIND 4 = before the short fast run d ^= a (KIND 0 then = before the long fast run c += d, b ^= c);  V 8: short fast run 1, long 0;  9: short 0, long 1
    if constexpr (V <= 7 && KIND == 4) pin_prio<V, 0>(a, b, c, d);
#define PP(v_, k_, txt) if constexpr (V == v_ && KIND == k_) asm volatile(txt : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
    PP(5, 0, "s_setprio 0"); PP(5, 1, "s_setprio 2"); PP(5, 2, "s_setprio 2"); PP(5, 3, "s_setprio 3");
    PP(6, 0, "s_setprio 0"); PP(6, 1, "s_setprio 3"); PP(6, 2, "s_setprio 3"); PP(6, 3, "s_setprio 2");
    PP(7, 0, "s_setprio 1"); PP(7, 1, "s_setprio 2"); PP(7, 2, ""); PP(7, 3, "s_setprio 2");
    PP(8, 0, "s_setprio 0"); PP(8, 4, "s_setprio 1"); PP(8, 1, "s_setprio 2"); PP(8, 2, "s_setprio 2"); PP(8, 3, "s_setprio 3");
    PP(9, 0, "s_setprio 1"); PP(9, 4, "s_setprio 0"); PP(9, 1, "s_setprio 2"); PP(9, 2, "s_setprio 2"); PP(9, 3, "s_setprio 3");
#undef PP
}
template <int LEAF, int V>
__device__ __forceinline__ void half_round_prio(uint32_t (&v)[16], const uint32_t (&m)[16], const int (&ia)[4], const int (&ib)[4], const int (&ic)[4],
                                                const int (&id)[4], const uint8_t* sx) {
#define PQ(KIND, ix) pin_prio<V, KIND>(v[ix[0]], v[ix[1]], v[ix[2]], v[ix[3]])
#define FQ for (int q = 0; q < 4; q++)
    _Pragma("unroll") FQ {
        const int x = sx[2 * q];
        v[ia[q]] = v[ia[q]] + v[ib[q]] + ((LEAF && x >= 4) ? 0u : m[x]);
    }
    PQ(4, ia);
    _Pragma("unroll") FQ v[id[q]] ^= v[ia[q]];
    PQ(3, id);
    _Pragma("unroll") FQ v[id[q]] = b2detail::rotr(v[id[q]], 16);
    PQ(0, id);
    _Pragma("unroll") FQ v[ic[q]] += v[id[q]];
    _Pragma("unroll") FQ v[ib[q]] ^= v[ic[q]];
    PQ(1, ib);
    _Pragma("unroll") FQ v[ib[q]] = b2detail::rotr(v[ib[q]], 12);
    _Pragma("unroll") FQ {
        const int y = sx[2 * q + 1];
        v[ia[q]] = v[ia[q]] + v[ib[q]] + ((LEAF && y >= 4) ? 0u : m[y]);
    }
    PQ(4, ia);
    _Pragma("unroll") FQ v[id[q]] ^= v[ia[q]];
    PQ(3, id);
    _Pragma("unroll") FQ v[id[q]] = b2detail::rotr(v[id[q]], 8);
    PQ(0, id);
    _Pragma("unroll") FQ v[ic[q]] += v[id[q]];
    _Pragma("unroll") FQ v[ib[q]] ^= v[ic[q]];
    PQ(3, ib);
    _Pragma("unroll") FQ v[ib[q]] = b2detail::rotr(v[ib[q]], 7);
    PQ(2, ib);
#undef FQ
#undef PQ
}
template <int LEAF, int WAVES, int V>
__global__ __launch_bounds__(256, WAVES) void chain_prio_kernel(uint32_t* out, int iters, Stamp* st) {
    using b2detail::IV;
    using b2detail::SIGMA;
    uint32_t m[16], h[8];
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int i = 0; i < 8; i++) h[i] = 0;
    unsigned long long c0, r0, c1, r1;
    stamp_pair(c0, r0);
    asm volatile("" : "+v"(m[0]) : "s"(c0));
    constexpr int ca[4] = {0, 1, 2, 3}, cb[4] = {4, 5, 6, 7}, cc[4] = {8, 9, 10, 11}, cd[4] = {12, 13, 14, 15};
    constexpr int db[4] = {5, 6, 7, 4}, dc[4] = {10, 11, 8, 9}, dd[4] = {15, 12, 13, 14};
    for (int it = 0; it < iters; it++) {
        uint32_t v[16];
        for (int i = 0; i < 8; i++) v[i] = 0u, v[8 + i] = IV[i];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            half_round_prio<LEAF, V>(v, m, ca, cb, cc, cd, &SIGMA[r][0]);
            half_round_prio<LEAF, V>(v, m, ca, db, dc, dd, &SIGMA[r][8]);
        }
        for (int i = 0; i < 8; i++) h[i] = v[i] ^ v[8 + i];
        if (LEAF) {
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    asm volatile("s_setprio 0" ::"v"(h[0]), "v"(h[7]));
    stamp_pair(c1, r1);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, r0, c1, r1};
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t*, int, Stamp*);

static double run(const char* name, kern_t kfn, int K, int waves, double seconds) {
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 256, 0);
    const int per_cu = std::min(occ, waves);
    const int blocks = 256 * per_cu, iters = 96;
    uint32_t* d_out;
    Stamp* d_st;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&d_st, sizeof(Stamp) * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; i++) kfn<<<blocks, 256>>>(d_out, iters, d_st);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(e0);
    kfn<<<blocks, 256>>>(d_out, iters, d_st);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(blocks);
    (void)hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (const Stamp& s : st) {
        const double dc = (double)(s.c1 - s.c0), dr = (double)(s.r1 - s.r0);
        if (dr > 0) clk.push_back(dc / dr * 0.1);
    }
    std::sort(clk.begin(), clk.end());
    const double clock = clk[clk.size() / 2];
    const double rate = (double)blocks * 256.0 * iters * K / (ms * 1e-3);
    uint32_t sum = 0;
    std::vector<uint32_t> ho((size_t)blocks * 256);
    (void)hipMemcpy(ho.data(), d_out, ho.size() * 4, hipMemcpyDeviceToHost);
    for (uint32_t x : ho) sum += x;
    printf("%-64s %d waves/SIMD  clock %5.3f GHz  %6.2f G compressions/s  %7.1f SIMD cycles per wave-compression  (checksum %08x)\n", name, per_cu, clock,
           rate * 1e-9, 1024.0 * 64.0 * clock * 1e9 / rate, sum);
    fflush(stdout);
    (void)hipFree(d_out);
    (void)hipFree(d_st);
    return rate;
}

#define RUN(K, LEAF, MODE, WAVES, label) run(label, chain_kernel<K, LEAF, MODE, WAVES>, K, WAVES, secs)

#define RUN_N(LEAF, WAVES, NA, NB, NC)                                                                                    \
    do {                                                                                                                  \
        char nm[96];                                                                                                      \
        snprintf(nm, sizeof nm, "%s K=1 idle states: slow>fast %d, fast>slow %d, slow>slow %d", LEAF ? "leaf" : "node", NA, NB, NC); \
        run(nm, chain_n_kernel<LEAF, WAVES, NA, NB, NC>, 1, WAVES, secs);                                                 \
    } while (0)

#define RUN_N2(LEAF, WAVES, NA, NB, NC)                                                                                   \
    do {                                                                                                                  \
        char nm[96];                                                                                                      \
        snprintf(nm, sizeof nm, "%s K=2 idle states: slow>fast %d, fast>slow %d, slow>slow %d", LEAF ? "leaf" : "node", NA, NB, NC); \
        run(nm, chain_n2_kernel<LEAF, WAVES, NA, NB, NC>, 2, WAVES, secs);                                                \
    } while (0)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.5;
    if (argc > 2 && argv[2][0] == 's') {  // wave priority switched at the run boundaries
        RUN_N(0, 8, 6, 0, 3); RUN_N(1, 8, 6, 0, 3);
        run("node prio: slow runs high, idle states 6-0-3", chain_prio_kernel<0, 8, 1>, 1, 8, secs);
        run("leaf prio: slow runs high, idle states 6-0-3", chain_prio_kernel<1, 8, 1>, 1, 8, secs);
        run("node prio: fast runs high, idle states 6-0-3", chain_prio_kernel<0, 8, 2>, 1, 8, secs);
        run("leaf prio: fast runs high, idle states 6-0-3", chain_prio_kernel<1, 8, 2>, 1, 8, secs);
        run("node prio: slow runs high, no idle states", chain_prio_kernel<0, 8, 3>, 1, 8, secs);
        run("leaf prio: slow runs high, no idle states", chain_prio_kernel<1, 8, 3>, 1, 8, secs);
        run("node prio: fast runs high, no idle states", chain_prio_kernel<0, 8, 4>, 1, 8, secs);
        run("leaf prio: fast runs high, no idle states", chain_prio_kernel<1, 8, 4>, 1, 8, secs);
        run("node prio: rotate runs 3, add3 runs 2, fast 0", chain_prio_kernel<0, 8, 5>, 1, 8, secs);
        run("leaf prio: rotate runs 3, add3 runs 2, fast 0", chain_prio_kernel<1, 8, 5>, 1, 8, secs);
        run("node prio: rotate runs 2, add3 runs 3, fast 0", chain_prio_kernel<0, 8, 6>, 1, 8, secs);
        run("leaf prio: rotate runs 2, add3 runs 3, fast 0", chain_prio_kernel<1, 8, 6>, 1, 8, secs);
        run("node prio: rotates 3, add3 2, d ^= a runs 1, c += d / b ^= c runs 0", chain_prio_kernel<0, 8, 8>, 1, 8, secs);
        run("leaf prio: rotates 3, add3 2, d ^= a runs 1, c += d / b ^= c runs 0", chain_prio_kernel<1, 8, 8>, 1, 8, secs);
        run("node prio: rotates 3, add3 2, d ^= a runs 0, c += d / b ^= c runs 1", chain_prio_kernel<0, 8, 9>, 1, 8, secs);
        run("leaf prio: rotates 3, add3 2, d ^= a runs 0, c += d / b ^= c runs 1", chain_prio_kernel<1, 8, 9>, 1, 8, secs);
        run("node prio: slow 2, fast 1", chain_prio_kernel<0, 8, 7>, 1, 8, secs);
        run("leaf prio: slow 2, fast 1", chain_prio_kernel<1, 8, 7>, 1, 8, secs);
        run("node prio: slow runs high, no idle states, 6 waves", chain_prio_kernel<0, 6, 3>, 1, 6, secs);
        run("leaf prio: slow runs high, no idle states, 6 waves", chain_prio_kernel<1, 6, 3>, 1, 6, secs);
        run("node prio: slow runs high, no idle states, 4 waves", chain_prio_kernel<0, 4, 3>, 1, 4, secs);
        run("leaf prio: slow runs high, no idle states, 4 waves", chain_prio_kernel<1, 4, 3>, 1, 4, secs);
        run("node prio: slow runs high, no idle states, 2 waves", chain_prio_kernel<0, 2, 3>, 1, 2, secs);
        run("leaf prio: slow runs high, no idle states, 2 waves", chain_prio_kernel<1, 2, 3>, 1, 2, secs);
        run("node prio: slow runs high, idle states, 4 waves", chain_prio_kernel<0, 4, 1>, 1, 4, secs);
        run("leaf prio: slow runs high, idle states, 4 waves", chain_prio_kernel<1, 4, 1>, 1, 4, secs);
        run("node prio: fast runs high, idle states, 4 waves", chain_prio_kernel<0, 4, 2>, 1, 4, secs);
        run("leaf prio: fast runs high, idle states, 4 waves", chain_prio_kernel<1, 4, 2>, 1, 4, secs);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'p') {  // per-position idle states, one coordinate at a time around 6-0-6-0-6-0-6-0-3
#define RUN_P(W, ...)                                                                              \
    do {                                                                                           \
        run("node K=1 per position " #__VA_ARGS__, chain_p_kernel<0, W, __VA_ARGS__>, 1, W, secs); \
        run("leaf K=1 per position " #__VA_ARGS__, chain_p_kernel<1, W, __VA_ARGS__>, 1, W, secs); \
    } while (0)
        RUN_P(8, 6, 0, 6, 0, 6, 0, 6, 0, 3);
        RUN_P(8, 4, 0, 6, 0, 6, 0, 6, 0, 3); RUN_P(8, 8, 0, 6, 0, 6, 0, 6, 0, 3); RUN_P(8, 3, 0, 6, 0, 6, 0, 6, 0, 3);
        RUN_P(8, 6, 0, 4, 0, 6, 0, 6, 0, 3); RUN_P(8, 6, 0, 8, 0, 6, 0, 6, 0, 3); RUN_P(8, 6, 0, 3, 0, 6, 0, 6, 0, 3);
        RUN_P(8, 6, 0, 6, 0, 4, 0, 6, 0, 3); RUN_P(8, 6, 0, 6, 0, 8, 0, 6, 0, 3); RUN_P(8, 6, 0, 6, 0, 3, 0, 6, 0, 3);
        RUN_P(8, 6, 0, 6, 0, 6, 0, 4, 0, 3); RUN_P(8, 6, 0, 6, 0, 6, 0, 8, 0, 3); RUN_P(8, 6, 0, 6, 0, 6, 0, 3, 0, 3);
        RUN_P(8, 6, 1, 6, 0, 6, 0, 6, 0, 3); RUN_P(8, 6, 0, 6, 1, 6, 0, 6, 0, 3); RUN_P(8, 6, 0, 6, 0, 6, 1, 6, 0, 3); RUN_P(8, 6, 0, 6, 0, 6, 0, 6, 1, 3);
        RUN_P(8, 6, 0, 6, 2, 6, 0, 6, 2, 3); RUN_P(8, 6, 0, 6, 0, 6, 0, 6, 0, 0); RUN_P(8, 0, 0, 6, 0, 6, 0, 6, 0, 3); RUN_P(8, 6, 0, 0, 0, 6, 0, 6, 0, 3);
        RUN_P(8, 6, 0, 6, 0, 0, 0, 6, 0, 3); RUN_P(8, 6, 0, 6, 0, 6, 0, 0, 0, 3);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'g') {  // around 6-0-3
#define BOTH(A, B, C, W) RUN_N(0, W, A, B, C); RUN_N(1, W, A, B, C)
        BOTH(6, 0, 3, 8); BOTH(6, 0, 2, 8); BOTH(6, 0, 4, 8); BOTH(6, 0, 5, 8); BOTH(5, 0, 3, 8); BOTH(7, 0, 3, 8); BOTH(8, 0, 3, 8); BOTH(8, 0, 4, 8);
        BOTH(7, 0, 4, 8); BOTH(10, 0, 3, 8); BOTH(10, 0, 5, 8); BOTH(12, 0, 6, 8); BOTH(6, 1, 3, 8); BOTH(7, 1, 3, 8); BOTH(5, 0, 2, 8); BOTH(5, 0, 4, 8);
        BOTH(6, 0, 3, 4); BOTH(8, 0, 4, 4); BOTH(4, 0, 2, 4); BOTH(3, 3, 3, 4); BOTH(5, 0, 3, 4); BOTH(3, 0, 2, 4);
        return 0;
    }
    if (argc > 2 && argv[2][0] == 'f') {  // finer sweep around the two optima of the first one, 8 waves per SIMD
        RUN_N(0, 8, 6, 0, 0); RUN_N(0, 8, 5, 0, 0); RUN_N(0, 8, 4, 0, 0); RUN_N(0, 8, 6, 1, 0); RUN_N(0, 8, 6, 0, 1); RUN_N(0, 8, 6, 0, 3); RUN_N(0, 8, 6, 0, 6);
        RUN_N(0, 8, 5, 1, 1); RUN_N(0, 8, 5, 2, 0); RUN_N(0, 8, 6, 2, 0); RUN_N(0, 8, 4, 2, 0); RUN_N(0, 8, 2, 4, 2); RUN_N(0, 8, 2, 5, 2); RUN_N(0, 8, 2, 6, 2);
        RUN_N(0, 8, 1, 4, 1); RUN_N(0, 8, 2, 4, 0); RUN_N(0, 8, 0, 4, 2); RUN_N(0, 8, 3, 3, 3); RUN_N(0, 8, 6, 0, 0);
        RUN_N(1, 8, 6, 0, 0); RUN_N(1, 8, 5, 0, 0); RUN_N(1, 8, 4, 0, 0); RUN_N(1, 8, 6, 1, 0); RUN_N(1, 8, 6, 0, 1); RUN_N(1, 8, 6, 0, 3); RUN_N(1, 8, 6, 0, 6);
        RUN_N(1, 8, 5, 1, 1); RUN_N(1, 8, 5, 2, 0); RUN_N(1, 8, 6, 2, 0); RUN_N(1, 8, 4, 2, 0); RUN_N(1, 8, 2, 4, 2); RUN_N(1, 8, 2, 5, 2); RUN_N(1, 8, 2, 6, 2);
        RUN_N(1, 8, 1, 4, 1); RUN_N(1, 8, 2, 4, 0); RUN_N(1, 8, 0, 4, 2); RUN_N(1, 8, 3, 3, 3); RUN_N(1, 8, 2, 4, 2);
        return 0;
    }
    if (argc > 2 && argv[2][0] == '2') {  // two compressions in lock step, at 4 and 8 waves per SIMD
        RUN_N(0, 4, 3, 3, 3); RUN_N(1, 4, 3, 3, 3); RUN_N(0, 4, 0, 0, 0); RUN_N(1, 4, 0, 0, 0);
        RUN_N2(0, 4, 1, 1, 1); RUN_N2(0, 4, 2, 2, 2); RUN_N2(0, 4, 3, 3, 3); RUN_N2(0, 4, 4, 4, 4); RUN_N2(0, 4, 3, 0, 3); RUN_N2(0, 4, 6, 0, 0);
        RUN_N2(1, 4, 1, 1, 1); RUN_N2(1, 4, 2, 2, 2); RUN_N2(1, 4, 3, 3, 3); RUN_N2(1, 4, 4, 4, 4); RUN_N2(1, 4, 3, 0, 3); RUN_N2(1, 4, 6, 0, 0);
        RUN_N2(0, 8, 3, 3, 3); RUN_N2(1, 8, 3, 3, 3); RUN_N2(0, 8, 5, 5, 5); RUN_N2(1, 8, 5, 5, 5);
        return 0;
    }
    if (argc > 2) {  // the idle-state sweep only
        for (int leaf = 0; leaf < 2; leaf++) {
            if (leaf == 0) {
                RUN_N(0, 8, 3, 3, 3); RUN_N(0, 8, 4, 4, 4); RUN_N(0, 8, 3, 3, 0); RUN_N(0, 8, 3, 0, 0); RUN_N(0, 8, 0, 3, 0); RUN_N(0, 8, 3, 0, 3);
                RUN_N(0, 8, 4, 2, 2); RUN_N(0, 8, 2, 4, 2); RUN_N(0, 8, 4, 3, 3); RUN_N(0, 8, 3, 4, 3); RUN_N(0, 8, 5, 3, 3); RUN_N(0, 8, 3, 2, 3);
                RUN_N(0, 8, 6, 0, 0); RUN_N(0, 8, 0, 6, 0); RUN_N(0, 8, 2, 2, 2); RUN_N(0, 8, 1, 1, 1); RUN_N(0, 8, 0, 0, 0);
                RUN_N(0, 4, 3, 3, 3); RUN_N(0, 4, 5, 5, 5); RUN_N(0, 4, 2, 2, 2);
            } else {
                RUN_N(1, 8, 3, 3, 3); RUN_N(1, 8, 4, 4, 4); RUN_N(1, 8, 3, 3, 0); RUN_N(1, 8, 3, 0, 0); RUN_N(1, 8, 0, 3, 0); RUN_N(1, 8, 3, 0, 3);
                RUN_N(1, 8, 4, 2, 2); RUN_N(1, 8, 2, 4, 2); RUN_N(1, 8, 4, 3, 3); RUN_N(1, 8, 3, 4, 3); RUN_N(1, 8, 5, 3, 3); RUN_N(1, 8, 3, 2, 3);
                RUN_N(1, 8, 6, 0, 0); RUN_N(1, 8, 0, 6, 0); RUN_N(1, 8, 2, 2, 2); RUN_N(1, 8, 1, 1, 1); RUN_N(1, 8, 0, 0, 0);
                RUN_N(1, 4, 3, 3, 3); RUN_N(1, 4, 5, 5, 5); RUN_N(1, 4, 2, 2, 2);
            }
        }
        return 0;
    }
    // node-shaped
    RUN(1, 0, 0, 8, "node K=1 product order (compiler schedule)");
    RUN(1, 0, 1, 8, "node K=1 lock step, no barriers");
    RUN(1, 0, 2, 8, "node K=1 runs of 4 (scheduling barriers)");
    RUN(2, 0, 0, 8, "node K=2 product order, one after another");
    RUN(2, 0, 1, 8, "node K=2 lock step, no barriers");
    RUN(2, 0, 2, 8, "node K=2 runs of 8 - 16");
    RUN(2, 0, 2, 4, "node K=2 runs of 8 - 16, 4 waves/SIMD");
    RUN(4, 0, 0, 4, "node K=4 product order, one after another");
    RUN(4, 0, 1, 4, "node K=4 lock step, no barriers");
    RUN(4, 0, 2, 4, "node K=4 runs of 16 - 32");
    RUN(4, 0, 2, 3, "node K=4 runs of 16 - 32, 3 waves/SIMD");
    RUN(1, 0, 3, 8, "node K=1 runs of 4, pinned by data flow");
    RUN(2, 0, 3, 8, "node K=2 runs of 8 - 16, pinned by data flow");
    RUN(2, 0, 3, 4, "node K=2 runs of 8 - 16, pinned, 4 waves/SIMD");
    RUN(1, 0, 4, 8, "node K=1 pinned + s_nop 0");
    RUN(1, 0, 5, 8, "node K=1 pinned + s_nop 1");
    RUN(1, 0, 6, 8, "node K=1 pinned + s_nop 3");
    RUN(1, 0, 7, 8, "node K=1 pinned in pairs (a nop every 2 instructions)");
    RUN(1, 0, 3, 4, "node K=1 pinned, 4 waves/SIMD");
    RUN(1, 0, 3, 2, "node K=1 pinned, 2 waves/SIMD");
    RUN(1, 0, 0, 2, "node K=1 product order, 2 waves/SIMD");
    // leaf-shaped
    RUN(1, 1, 4, 8, "leaf K=1 pinned + s_nop 0");
    RUN(1, 1, 5, 8, "leaf K=1 pinned + s_nop 1");
    RUN(1, 1, 6, 8, "leaf K=1 pinned + s_nop 3");
    RUN(1, 1, 7, 8, "leaf K=1 pinned in pairs (a nop every 2 instructions)");
    RUN(1, 1, 3, 4, "leaf K=1 pinned, 4 waves/SIMD");
    RUN(1, 1, 3, 8, "leaf K=1 runs of 4, pinned by data flow");
    RUN(2, 1, 3, 8, "leaf K=2 runs of 8 - 16, pinned by data flow");
    RUN(4, 1, 3, 4, "leaf K=4 runs of 16 - 32, pinned by data flow");
    RUN(4, 1, 3, 6, "leaf K=4 runs of 16 - 32, pinned, <= 6 waves/SIMD");
    RUN(1, 1, 0, 8, "leaf K=1 product order (compiler schedule)");
    RUN(1, 1, 2, 8, "leaf K=1 runs of 4");
    RUN(2, 1, 2, 8, "leaf K=2 runs of 8 - 16");
    RUN(4, 1, 0, 4, "leaf K=4 product order, one after another");
    RUN(4, 1, 2, 4, "leaf K=4 runs of 16 - 32");
    RUN(4, 1, 2, 6, "leaf K=4 runs of 16 - 32, <= 6 waves/SIMD");
    RUN(8, 1, 2, 3, "leaf K=8 runs of 32 - 64");
    return 0;
}
