// intt.hip — the reconstruction side: inverse circle FFT of one aligned block of the codeword
// and the 30-bit packer (gfx950).
//
// Block k of the bit-reversed evaluation (entries k * 2^L .. (k+1) * 2^L of a column, any k < 2^(n-L)) holds the values of
// the polynomial on one sub-coset of the domain; the encode produced it with the L layers i = L-1 .. 0 and the twiddles
// T_{i-1}[(k << (L-1-i)) | h] (ntt.hip).  Undoing those layers in the opposite order with the inverse twiddles and scaling
// by 2^-L therefore recovers the 2^L coefficients from ANY 1 / 2^B of the codeword.  With k = 0 and L = n this is stwo's
// `CpuBackend::interpolate` (backend/cpu/circle.rs; core/fft.rs::ibutterfly) on the canonic domain.  frieda itself never
// calls interpolate (its README's `sample()` / reconstruction API is not in /root/reference/src), so parity here is against
// the oracle's restatement and the round trip evaluate -> block -> interpolate == identity.
//
// Passes mirror the encode: the first pass takes the low (contiguous) layers, later passes the strided ones; inside a pass a
// thread runs up to four layers on 2^R elements in registers, lowest layer first.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace frieda {
namespace k {

namespace {

constexpr int INTT_THREADS = 256;
constexpr uint32_t ITILE_LOG = 12;
constexpr uint32_t ITILE_WORDS = (1u << ITILE_LOG) + (1u << (ITILE_LOG - 4));
constexpr uint32_t IMID_LOG_W = 4;

__device__ __forceinline__ uint32_t ipad(uint32_t e) { return e + (e >> 4); }

__device__ __forceinline__ uint32_t inv_circle_tw(const uint32_t* __restrict__ itw, uint32_t n, uint32_t h, uint32_t inv_init_y) {
    if (n < 3) return (h & 1u) ? m31_neg(inv_init_y) : inv_init_y;
    uint32_t j = h >> 2, r = h & 3u;
    uint32_t v = itw[2 * j + (r < 2 ? 1 : 0)];
    return (r == 1 || r == 2) ? m31_neg(v) : v;
}

struct InttArgs {
    const uint32_t* in;
    size_t in_stride;
    uint32_t* out;
    size_t out_stride;
    const uint32_t* itw;
    uint32_t n, L, block;     // domain log, coefficient log, block index k
    uint32_t i_hi, i_lo, log_w, inv_init_y;
    uint32_t n_stages;
    uint32_t stage_r[3];  // layers per stage, LOWEST stage first
    uint32_t scale;       // multiply every output by this (2^-L on the last pass, 1 otherwise)
};

template <int R>
__device__ __forceinline__ void inv_stage(uint32_t* lds, const InttArgs& a, uint32_t tb, uint32_t lo, uint32_t hblk) {
    constexpr int E = 1 << R;
    const uint32_t n_groups = 1u << (tb - R);
    for (uint32_t g = threadIdx.x; g < n_groups; g += INTT_THREADS) {
        const uint32_t base = ((g >> lo) << (lo + R)) | (g & ((1u << lo) - 1));
        uint32_t x[E];
#pragma unroll
        for (int r = 0; r < E; r++) x[r] = lds[ipad(base | ((uint32_t)r << lo))];
        // layers ascending: stage bit `bit` = 0 .. R-1 (tile bit b = lo + bit, global layer i = i_lo + b - log_w)
#pragma unroll
        for (int bit = 0; bit < R; bit++) {
            const uint32_t b = lo + bit;
            const uint32_t i = a.i_lo + b - a.log_w;
            const uint32_t hbase = (hblk << (a.i_hi - i)) | (base >> (b + 1));
#pragma unroll
            for (int r = 0; r < E; r++) {
                if (r & (1 << bit)) continue;
                const uint32_t h = hbase | (uint32_t)(r >> (bit + 1));
                const uint32_t t = (i >= 1) ? a.itw[tw_level_offset_dev(a.n, i - 1) + h] : inv_circle_tw(a.itw, a.n, h, a.inv_init_y);
                const uint32_t v0 = x[r], v1 = x[r | (1 << bit)];
                x[r] = m31_add(v0, v1);
                x[r | (1 << bit)] = m31_mul(m31_sub(v0, v1), t);
            }
        }
#pragma unroll
        for (int r = 0; r < E; r++) lds[ipad(base | ((uint32_t)r << lo))] = x[r];
    }
}

__global__ __launch_bounds__(INTT_THREADS) void intt_tile_kernel(InttArgs a) {
    __shared__ uint32_t lds[ITILE_WORDS];
    const uint32_t t = a.i_hi - a.i_lo + 1;
    const uint32_t tb = t + a.log_w;
    const uint32_t tile = 1u << tb;
    const uint32_t wmask = (1u << a.log_w) - 1;
    const uint32_t nwb_log = a.i_lo - a.log_w;
    const uint32_t wblk = blockIdx.x & ((1u << nwb_log) - 1);
    const uint32_t hloc = blockIdx.x >> nwb_log;                   // index bits above i_hi inside the block
    const uint32_t hblk = (a.block << (a.L - 1 - a.i_hi)) | hloc;  // ... and the block number above those
    const uint32_t gbase = (hloc << (a.i_hi + 1)) | (wblk << a.log_w);
    const uint32_t* in = a.in + (size_t)blockIdx.y * a.in_stride;
    uint32_t* out = a.out + (size_t)blockIdx.y * a.out_stride;

    for (uint32_t e = threadIdx.x; e < tile; e += INTT_THREADS) {
        uint32_t g = gbase | ((e >> a.log_w) << a.i_lo) | (e & wmask);
        lds[ipad(e)] = in[g];
    }
    __syncthreads();
    uint32_t lo = a.log_w;
    for (uint32_t s = 0; s < a.n_stages; s++) {
        const uint32_t r = a.stage_r[s];
        switch (r) {
            case 4: inv_stage<4>(lds, a, tb, lo, hblk); break;
            case 3: inv_stage<3>(lds, a, tb, lo, hblk); break;
            case 2: inv_stage<2>(lds, a, tb, lo, hblk); break;
            default: inv_stage<1>(lds, a, tb, lo, hblk); break;
        }
        lo += r;
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < tile; e += INTT_THREADS) {
        uint32_t g = gbase | ((e >> a.log_w) << a.i_lo) | (e & wmask);
        uint32_t v = lds[ipad(e)];
        out[g] = a.scale == 1u ? v : m31_mul(v, a.scale);
    }
}

void set_inv_stages(InttArgs& a, uint32_t t) {
    uint32_t ns = (t + 3) / 4;
    a.n_stages = ns;
    uint32_t left = t;
    for (uint32_t s = 0; s < ns; s++) {
        uint32_t r = (left + (ns - s) - 1) / (ns - s);
        a.stage_r[s] = r;
        left -= r;
    }
}

// felts -> bytes: output dword d holds stream bits [32 d, 32 d + 32), i.e. pieces of the felts floor(32 d / 30) ..
__global__ __launch_bounds__(256) void pack30_kernel(const uint32_t* __restrict__ felts, size_t n_felts, uint8_t* __restrict__ out,
                                                     size_t len) {
    size_t d = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t n_dw = (len + 3) / 4;
    if (d >= n_dw) return;
    const size_t bit0 = 32 * d;
    size_t kf = bit0 / 30;
    uint32_t off = (uint32_t)(bit0 - 30 * kf);  // bit offset inside felt kf where this dword starts
    uint64_t acc = 0;
    uint32_t have = 0;
    while (have < 32) {
        uint64_t f = kf < n_felts ? (uint64_t)(felts[kf] & 0x3fffffffu) : 0ull;
        acc |= (f >> off) << have;
        have += 30 - off;
        off = 0;
        kf++;
    }
    uint32_t w = (uint32_t)acc;
    size_t b = 4 * d;
    if (b + 4 <= len && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(out + b) = w;
    } else {
        for (int i = 0; i < 4 && b + i < len; i++) out[b + i] = (uint8_t)(w >> (8 * i));
    }
}

}  // namespace

void circle_interpolate_block(const Launch& L_, const uint32_t* d_block, size_t in_stride, uint32_t ncols, uint32_t L, uint32_t n,
                              uint32_t block, const uint32_t* d_itw, DomainScalars ds, uint32_t* d_coef, size_t out_stride) {
    hipStream_t s = L_.stream;
    const size_t M = (size_t)1 << L;
    Scope scope(L_, "intt_block", 8.0 * ncols * (double)M);
    if (L == 0) {
        (void)hipMemcpy2DAsync(d_coef, out_stride * 4, d_block, in_stride * 4, 4, ncols, hipMemcpyDeviceToDevice, s);
        return;
    }
    InttArgs a{};
    a.in = d_block;
    a.in_stride = in_stride;
    a.out = d_coef;
    a.out_stride = out_stride;
    a.itw = d_itw;
    a.n = n;
    a.L = L;
    a.block = block;
    a.inv_init_y = ds.inv_init_y;
    const uint32_t scale = (1u << (31 - L)) % P31;  // 2^-L = 2^(31-L) mod P
    // first pass: layers 0 .. t0-1 (contiguous); later passes: up to 8 strided layers each
    const uint32_t t0 = L < ITILE_LOG ? L : ITILE_LOG;
    uint32_t done = 0;
    {
        a.i_lo = 0;
        a.i_hi = t0 - 1;
        a.log_w = 0;
        set_inv_stages(a, t0);
        done = t0;
        a.scale = done == L ? scale : 1u;
        dim3 grid((unsigned)(M >> t0), ncols);
        intt_tile_kernel<<<grid, INTT_THREADS, 0, s>>>(a);
        a.in = d_coef;
        a.in_stride = out_stride;
    }
    while (done < L) {
        const uint32_t mid_max = ITILE_LOG - IMID_LOG_W;
        uint32_t t = L - done < mid_max ? L - done : mid_max;
        a.i_lo = done;
        a.i_hi = done + t - 1;
        a.log_w = IMID_LOG_W;  // done >= 12 here
        set_inv_stages(a, t);
        done += t;
        a.scale = done == L ? scale : 1u;
        dim3 grid((unsigned)(M >> (t + a.log_w)), ncols);
        intt_tile_kernel<<<grid, INTT_THREADS, 0, s>>>(a);
    }
}

// coef[col][u * M + t] = sum_r vinv[u][r] * w[r][col][t]   (M = 2^m words per cell and column, R cells; see cells_combine below)
namespace {
__global__ __launch_bounds__(256) void cells_combine_kernel(const uint32_t* __restrict__ w, const uint32_t* __restrict__ vinv, uint32_t R,
                                                            uint32_t ncols, size_t M, uint32_t* __restrict__ coef, size_t coef_stride) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t u = blockIdx.y, col = blockIdx.z;
    if (t >= M) return;
    const uint32_t* row = vinv + (size_t)u * R;
    uint64_t acc = 0;
    uint32_t pending = 0;
    for (uint32_t r = 0; r < R; r++) {
        acc += (uint64_t)row[r] * w[((size_t)r * ncols + col) * M + t];  // each product < 2^62
        if (++pending == 3) {  // three products + a reduced value stay below 2^64
            acc = m31_reduce64(acc);
            pending = 0;
        }
    }
    coef[(size_t)col * coef_stride + (size_t)u * M + t] = m31_reduce64(acc);
}
}  // namespace

// Second half of the reconstruction from scattered cells (oracle: fo_reconstruct_cells): d_w[R][ncols][2^m] holds the cells after
// their block transforms were undone, d_vinv[R][R] the inverse of the matrix V[c][u] = prod_{b in u} s_b(c); the coefficient
// slice u (entries u * 2^m .. of every column) is the combination sum_r vinv[u][r] * w[r].
void cells_combine(const Launch& L_, const uint32_t* d_w, const uint32_t* d_vinv, uint32_t R, uint32_t ncols, uint32_t m, uint32_t* d_coef,
                   size_t coef_stride) {
    const size_t M = (size_t)1 << m;
    Scope scope(L_, "cells_combine", 8.0 * ncols * (double)R * (double)M);
    dim3 grid((unsigned)((M + 255) / 256), R, ncols);
    cells_combine_kernel<<<grid, 256, 0, L_.stream>>>(d_w, d_vinv, R, ncols, M, d_coef, coef_stride);
}

void pack30(const Launch& L_, const uint32_t* d_felts, size_t n_felts, uint8_t* d_out, size_t len) {
    if (len == 0) return;
    Scope scope(L_, "pack30", 4.0 * (double)n_felts + (double)len);
    size_t n_dw = (len + 3) / 4;
    pack30_kernel<<<(unsigned)((n_dw + 255) / 256), 256, 0, L_.stream>>>(d_felts, n_felts, d_out, len);
}

}  // namespace k
}  // namespace frieda
