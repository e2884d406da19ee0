// twiddle.hip — twiddle / inverse-twiddle tables for the circle FFT and the FRI folds (gfx950).
//
// Replaces `CpuBackend::precompute_twiddles(Coset::half_odds(n - 1))` (/root/reference/src/commit.rs:15,
// src/proof.rs:47; stwo backend/cpu/circle.rs).  Table layout is stwo's: levels of size N/4, N/8, ..., 1
// followed by the pad value 1, 2^(n-1) words in all; level l holds the x-coordinates of the first half of
// the l-times-doubled coset in bit-reversed order.
//
// The reference walks the coset by repeated point addition and bit-reverses afterwards.  Here every thread
// owns one level-0 entry h: it builds point(i0 + brev(h) * step) from a 32-entry table of step * 2^k points
// (<= n-2 point additions), and then follows the doubling chain x -> 2x^2 - 1 upwards, because
// T_{l+1}[h] = double_x(T_l[2h]): a thread whose index has z trailing zeros also produces levels 1..z.
// Inverses come from the fixed 37-multiplication addition chain for x^(P-2).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace frieda {
namespace k {

namespace {

__global__ __launch_bounds__(256) void gen_twiddles_kernel(uint32_t n, TwiddleSeeds seeds, uint32_t* __restrict__ tw,
                                                           uint32_t* __restrict__ itw) {
    // n >= 3: level 0 has 2^(n-2) entries
    uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 1u << (n - 2);
    if (h >= cnt) return;
    uint32_t j = bit_reverse(h, n - 2);
    CPoint p = seeds.p0;
    for (uint32_t b = 0; b < n - 2; b++)
        if ((j >> b) & 1u) p = cp_add(p, seeds.step[b]);
    uint32_t x = p.x;
    tw[h] = x;
    itw[h] = m31_inv(x);
    uint32_t hh = h;
    for (uint32_t lv = 1; lv <= n - 2; lv++) {
        if (hh & 1u) break;
        hh >>= 1;
        x = double_x(x);
        size_t o = tw_level_offset_dev(n, lv) + hh;
        tw[o] = x;
        itw[o] = m31_inv(x);
    }
    if (h == 0) {
        size_t last = ((size_t)1 << (n - 1)) - 1;
        tw[last] = 1;
        itw[last] = 1;
    }
}

__global__ void gen_twiddles_tiny_kernel(uint32_t n, TwiddleSeeds seeds, uint32_t* tw, uint32_t* itw) {
    // n == 1: [pad]; n == 2: [x(initial), pad]
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n == 2) {
        tw[0] = seeds.p0.x;
        itw[0] = m31_inv(seeds.p0.x);
        tw[1] = 1;
        itw[1] = 1;
    } else {
        tw[0] = 1;
        itw[0] = 1;
    }
}

}  // namespace

void gen_twiddles(const Launch& L, uint32_t n, const TwiddleSeeds& seeds, uint32_t* d_tw, uint32_t* d_itw) {
    hipStream_t s = L.stream;
    Scope scope(L, "gen_twiddles", 8.0 * (double)((size_t)1 << (n - 1)));
    if (n < 3) {
        gen_twiddles_tiny_kernel<<<1, 64, 0, s>>>(n, seeds, d_tw, d_itw);
        return;
    }
    uint32_t cnt = 1u << (n - 2);
    gen_twiddles_kernel<<<(cnt + 255) / 256, 256, 0, s>>>(n, seeds, d_tw, d_itw);
}

}  // namespace k
}  // namespace frieda
