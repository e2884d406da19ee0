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

// ---- fast path for n >= 12 ----
// Level-0 entry h = 1024 w + r (r < 1024) is x(P0 + brev(h) S) with brev(h, n-2) = brev(r, 10) << (n-12) | brev(w, n-12):
// a workgroup-constant point Q_w = P0 + brev(w) S plus one of 1024 table points T10[r'] = r' (2^(n-12) S).  So an entry costs
// one x-coordinate of a point addition (2 multiplications) instead of ~(n-2)/2 point additions, and a thread that owns four
// consecutive entries inverts its seven chain values (4 + 2 + 1 over levels 0..2) with one shared field inversion.
__global__ __launch_bounds__(256) void twiddle_table_kernel(uint32_t n, TwiddleSeeds seeds, CPoint* __restrict__ t10) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 1024) return;
    CPoint p{1, 0};
    for (uint32_t b = 0; b < 10; b++)
        if ((r >> b) & 1u) p = cp_add(p, seeds.step[b + (n - 12)]);
    t10[r] = p;
}

__global__ __launch_bounds__(256) void gen_twiddles_fast_kernel(uint32_t n, TwiddleSeeds seeds, const CPoint* __restrict__ t10,
                                                                uint32_t* __restrict__ tw, uint32_t* __restrict__ itw) {
    __shared__ CPoint qw;
    const uint32_t w = blockIdx.x, tq = threadIdx.x;
    if (tq == 0) {
        uint32_t j = bit_reverse(w, n - 12);
        CPoint p = seeds.p0;
        for (uint32_t b = 0; b + 12 < n; b++)
            if ((j >> b) & 1u) p = cp_add(p, seeds.step[b]);
        qw = p;
    }
    __syncthreads();
    const CPoint q = qw;
    const uint32_t h = 1024u * w + 4u * tq;
    uint32_t v[7];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const CPoint tp = t10[bit_reverse(4u * tq + (uint32_t)e, 10)];
        v[e] = m31_sub(m31_mul(q.x, tp.x), m31_mul(q.y, tp.y));  // x of q + tp
    }
    v[4] = double_x(v[0]);
    v[5] = double_x(v[2]);
    v[6] = double_x(v[4]);
    // batch inversion (Montgomery's trick)
    uint32_t pre[7];
    pre[0] = v[0];
#pragma unroll
    for (int i = 1; i < 7; i++) pre[i] = m31_mul(pre[i - 1], v[i]);
    uint32_t inv = m31_inv(pre[6]);
    uint32_t iv[7];
#pragma unroll
    for (int i = 6; i >= 1; i--) {
        iv[i] = m31_mul(inv, pre[i - 1]);
        inv = m31_mul(inv, v[i]);
    }
    iv[0] = inv;
    reinterpret_cast<uint4*>(tw)[h >> 2] = make_uint4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<uint4*>(itw)[h >> 2] = make_uint4(iv[0], iv[1], iv[2], iv[3]);
    const size_t o1 = tw_level_offset_dev(n, 1) + (h >> 1), o2 = tw_level_offset_dev(n, 2) + (h >> 2);
    tw[o1] = v[4];
    tw[o1 + 1] = v[5];
    itw[o1] = iv[4];
    itw[o1 + 1] = iv[5];
    tw[o2] = v[6];
    itw[o2] = iv[6];
    // the rest of the doubling chain, for the entries whose index keeps being even
    uint32_t hh = h >> 2, x = v[6];
    for (uint32_t lv = 3; lv <= n - 2; lv++) {
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

void gen_twiddles(const Launch& L, uint32_t n, const TwiddleSeeds& seeds, uint32_t* d_tw, uint32_t* d_itw, void* d_scratch8k) {
    hipStream_t s = L.stream;
    Scope scope(L, "gen_twiddles", 8.0 * (double)((size_t)1 << (n - 1)));
    if (n < 3) {
        gen_twiddles_tiny_kernel<<<1, 64, 0, s>>>(n, seeds, d_tw, d_itw);
        return;
    }
    if (n >= 12 && d_scratch8k) {
        CPoint* t10 = static_cast<CPoint*>(d_scratch8k);
        twiddle_table_kernel<<<4, 256, 0, s>>>(n, seeds, t10);
        gen_twiddles_fast_kernel<<<1u << (n - 12), 256, 0, s>>>(n, seeds, t10, d_tw, d_itw);
        return;
    }
    uint32_t cnt = 1u << (n - 2);
    gen_twiddles_kernel<<<(cnt + 255) / 256, 256, 0, s>>>(n, seeds, d_tw, d_itw);
}

}  // namespace k
}  // namespace frieda
