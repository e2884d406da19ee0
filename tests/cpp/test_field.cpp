// Host check of the lazily reduced field kernels in frieda_amd/csrc/field.h against the textbook sequences
// (stwo core/fields/qm31.rs arithmetic; fold formula of stwo core/fri.rs::fold_line / fold_circle_into_line).
#include <cstdint>
#include <cstdio>

#include "field.h"

using namespace frieda;

static uint64_t sm_state = 1;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t pick() {
    static const uint32_t edge[] = {0u, 1u, 2u, P31 - 1, P31 - 2, 0x40000000u, 0x3fffffffu};
    uint64_t r = splitmix();
    if ((r & 7) < 3) return edge[(r >> 8) % 7];
    return (uint32_t)((r >> 16) % P31);
}

int main() {
    long bad = 0;
    // m31_reduce64 against the 128-bit remainder, including the extremes of the 64-bit range
    const uint64_t ext[] = {0ull, 1ull, P31, (uint64_t)P31 * P31, ~0ull, ~0ull - 1, 0x7fffffffffffffffull, 0x8000000000000000ull,
                            0xfffffffe00000001ull, (uint64_t)P31 << 31, ((uint64_t)P31 << 32) | P31};
    for (uint64_t v : ext)
        if (m31_reduce64(v) != (uint32_t)(v % P31)) bad++;
    for (int i = 0; i < 2000000; i++) {
        uint64_t v = splitmix();
        if (m31_reduce64(v) != (uint32_t)(v % P31)) bad++;
    }
    // qm_matrix: M(s) f == s * f
    for (int i = 0; i < 300000; i++) {
        QM31 s{pick(), pick(), pick(), pick()}, x{pick(), pick(), pick(), pick()}, y{pick(), pick(), pick(), pick()};
        uint32_t it = pick();
        QM31 f0 = qm_add(x, y), f1 = qm_scale(qm_sub(x, y), it);
        QM31 want = qm_add(f0, qm_mul(s, f1));
        QM31 got = qm_fold_pair(x, y, it, qm_matrix(s));
        if (!qm_eq(want, got)) bad++;
        if (got.a >= P31 || got.b >= P31 || got.c >= P31 || got.d >= P31) bad++;
    }
    // all-maximal operands: the accumulator bound 4 (P-1)^2 + 2(P-1) < 2^64 is attained here
    {
        const uint32_t m = P31 - 1;
        QM31 s{m, 1, m, 1}, x{m, m, m, m}, y{0, 0, 0, 0};  // matrix entries of s include P-1 in every row
        QM31 want = qm_add(qm_add(x, y), qm_mul(s, qm_scale(qm_sub(x, y), m)));
        if (!qm_eq(want, qm_fold_pair(x, y, m, qm_matrix(s)))) bad++;
    }
    printf("%s bad=%ld\n", bad ? "FAIL" : "OK", bad);
    return bad ? 1 : 0;
}
