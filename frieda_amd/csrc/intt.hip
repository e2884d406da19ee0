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

#include <cstdlib>

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
    uint32_t ncols;       // intt_tile12_kernel: columns handled by one workgroup (<= 4); grid.y strides over groups of this many
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

// ------------------------------------------------------------------------------------------------
// The hot shapes, mirror image of ntt_tile12_kernel (ntt.hip): a 4096-word tile whose layers split into NS stages of exactly four —
// NS = 3, LOG_W = 0: the contiguous first pass of 12 layers; NS = 2, LOG_W = 4: a strided pass of 8 layers (64-byte runs); NS = 1,
// LOG_W = 8: a strided pass of 4 layers (1 KiB runs).  One group of 16 elements per thread and stage, lowest stage first; the 15 * NS
// inverse twiddles of the thread are loaded once and serve the workgroup's (up to four) columns, those of the stage on tile bits 8..11
// are workgroup-uniform and live in scalar registers.  The first stage takes its 16 elements straight from memory (the contiguous pass:
// four 16-byte loads of the thread's own 16 consecutive words; the strided passes: 16 coalesced 4-byte loads), the last stage stores
// straight from registers (element g + 256 r: consecutive lanes, consecutive words), so the 17 KiB LDS tile only carries the exchange
// between stages — none at all for the 4-layer pass — and the next column's elements are prefetched while the current one is computed.
// ------------------------------------------------------------------------------------------------
template <int NS, int LOG_W>
__global__ __launch_bounds__(INTT_THREADS) void intt_tile12_kernel(InttArgs a) {
    __shared__ uint32_t lds[NS > 1 ? ITILE_WORDS : 1];
    const uint32_t g = threadIdx.x;
    const uint32_t nwb_log = a.i_lo - LOG_W;
    const uint32_t wblk = blockIdx.x & ((1u << nwb_log) - 1);
    const uint32_t hloc = blockIdx.x >> nwb_log;                   // index bits above i_hi inside the block
    const uint32_t hblk = (a.block << (a.L - 1 - a.i_hi)) | hloc;  // ... and the block number above those
    const uint32_t gbase = (hloc << (a.i_hi + 1)) | (wblk << LOG_W);
    constexpr uint32_t wmask = (1u << LOG_W) - 1;
    const size_t col0 = (size_t)blockIdx.y * a.ncols;
    const uint32_t* in = a.in + col0 * a.in_stride;
    uint32_t* out = a.out + col0 * a.out_stride;

    uint32_t pbase[NS];
    uint32_t twd[NS][15];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const uint32_t lo = LOG_W + 4u * (uint32_t)s;
        const uint32_t base = ((g >> lo) << (lo + 4)) | (g & ((1u << lo) - 1));
        pbase[s] = ipad(base);
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            const int q = 3 - bit;  // 2^q distinct twiddles in this layer of the group
            const uint32_t b = lo + (uint32_t)bit;
            const uint32_t i = a.i_lo + b - LOG_W;
            const uint32_t hbase = (hblk << (a.i_hi - i)) | (base >> (b + 1));  // its low q bits are zero
            if (LOG_W == 0 && s == 0 && bit == 0) {  // i == 0: the circle layer
#pragma unroll
                for (int u = 0; u < 8; u++) twd[s][7 + u] = inv_circle_tw(a.itw, a.n, hbase + (uint32_t)u, a.inv_init_y);
            } else {
                const uint32_t* lvl = a.itw + tw_level_offset_dev(a.n, i - 1) + hbase;
#pragma unroll
                for (int u = 0; u < (1 << q); u++) {
                    uint32_t v = lvl[u];
                    if (LOG_W + 4 * s == 8) v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);  // tile bits 8..11: no thread-dependent index bits
                    twd[s][(1 << q) - 1 + u] = v;
                }
            }
        }
    }

    // the thread's 16 elements of the first stage (tile bits LOG_W .. LOG_W + 3) and of the last one (tile bits 8 .. 11), as global indices
    auto global_of = [&](uint32_t e) { return gbase | ((e >> LOG_W) << a.i_lo) | (e & wmask); };
    const uint32_t base_first = ((g >> LOG_W) << (LOG_W + 4)) | (g & wmask);
    auto load16 = [&](const uint32_t* src, uint32_t* x) {
        if (LOG_W == 0) {
            const uint4* p = reinterpret_cast<const uint4*>(src + gbase + 16u * g);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint4 v = p[kk];
                x[4 * kk] = v.x;
                x[4 * kk + 1] = v.y;
                x[4 * kk + 2] = v.z;
                x[4 * kk + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = src[global_of(base_first | ((uint32_t)r << LOG_W))];
        }
    };

    uint32_t pre[16];
    load16(in, pre);
    for (uint32_t c = 0; c < a.ncols; c++) {
        uint32_t x[16];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = pre[r];
        if (c + 1 < a.ncols) load16(in + (size_t)(c + 1) * a.in_stride, pre);  // prefetch the next column
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t lo = LOG_W + 4u * (uint32_t)s;
            uint32_t* col = lds + pbase[s];
            if (s > 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) x[r] = col[ipad((uint32_t)r << lo)];
            }
#pragma unroll
            for (int bit = 0; bit < 4; bit++) {
                const int q = 3 - bit;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (r & (1 << bit)) continue;
                    const int u = r >> (bit + 1);
                    const uint32_t v0 = x[r], v1 = x[r | (1 << bit)];
                    x[r] = m31_add(v0, v1);
                    x[r | (1 << bit)] = m31_mul(m31_sub(v0, v1), twd[s][(1 << q) - 1 + u]);
                }
            }
            if (s + 1 < NS) {
#pragma unroll
                for (int r = 0; r < 16; r++) col[ipad((uint32_t)r << lo)] = x[r];
                // The stages on tile bits 0..3 and 4..7 only exchange data inside a wave's own 1024 elements (as in ntt_tile12_kernel): no
                // workgroup barrier between those two, only program order.
                if (LOG_W == 0 && NS == 3 && s == 0)
                    __builtin_amdgcn_wave_barrier();
                else
                    __syncthreads();
            }
        }
        // last stage: tile bits 8..11, the thread holds elements g + 256 r
        uint32_t* dst = out + (size_t)c * a.out_stride;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t v = a.scale == 1u ? x[r] : m31_mul(x[r], a.scale);
            dst[global_of(g | ((uint32_t)r << 8))] = v;
        }
        if (NS > 1 && c + 1 < a.ncols) __syncthreads();  // the next column's first exchange overwrites the tile other waves may still read
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
    // 12 or more layers over 16-byte aligned buffers: the first 12 layers, then the strided ones 8 and 4 at a time, as passes of
    // intt_tile12_kernel; what is left (L mod 4 layers) and every other shape goes through the generic kernel below.
    const bool no_fast = L_.tune->intt_generic;  // A/B knob
    const bool aligned = ((in_stride | out_stride) & 3) == 0 && ((reinterpret_cast<uintptr_t>(d_block) | reinterpret_cast<uintptr_t>(d_coef)) & 15) == 0;
    uint32_t done = 0;
    if (L >= ITILE_LOG && aligned && !no_fast) {
        uint32_t cpw = 4;
        while (ncols % cpw) cpw--;
        a.ncols = cpw;
        const dim3 grid((unsigned)(M >> ITILE_LOG), ncols / cpw);
        a.i_lo = 0;
        a.i_hi = ITILE_LOG - 1;
        a.scale = L == ITILE_LOG ? scale : 1u;
        intt_tile12_kernel<3, 0><<<grid, INTT_THREADS, 0, s>>>(a);
        done = ITILE_LOG;
        a.in = d_coef;
        a.in_stride = out_stride;
        while (L - done >= 4) {
            const uint32_t t = L - done >= 8 ? 8u : 4u;
            a.i_lo = done;
            a.i_hi = done + t - 1;
            done += t;
            a.scale = done == L ? scale : 1u;
            if (t == 8)
                intt_tile12_kernel<2, 4><<<grid, INTT_THREADS, 0, s>>>(a);
            else
                intt_tile12_kernel<1, 8><<<grid, INTT_THREADS, 0, s>>>(a);
        }
    }
    // first pass: layers 0 .. t0-1 (contiguous); later passes: up to 8 strided layers each
    const uint32_t t0 = L < ITILE_LOG ? L : ITILE_LOG;
    if (done == 0) {
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
        a.log_w = ITILE_LOG - t;  // done >= 12 >= log_w here: a pass of fewer than 8 layers takes longer runs so that its tile still has 4096 words
        set_inv_stages(a, t);
        done += t;
        a.scale = done == L ? scale : 1u;
        dim3 grid((unsigned)(M >> (t + a.log_w)), ncols);
        intt_tile_kernel<<<grid, INTT_THREADS, 0, s>>>(a);
    }
}

// coef[col][u * M + t] = sum_r vinv[u][r] * w[r][col][t]   (M = 2^m words per cell and column, R cells; see cells_combine below)
namespace {
__global__ __launch_bounds__(256) void cells_combine_kernel(const uint32_t* __restrict__ w, const uint32_t* __restrict__ vinv, size_t vinv_pitch,
                                                            uint32_t R, uint32_t ncols, size_t M, uint32_t* __restrict__ coef, size_t coef_stride) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t u = blockIdx.y, col = blockIdx.z;
    if (t >= M) return;
    const uint32_t* row = vinv + (size_t)u * vinv_pitch;
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
                   size_t coef_stride, size_t vinv_pitch) {
    if (vinv_pitch == 0) vinv_pitch = R;
    const size_t M = (size_t)1 << m;
    Scope scope(L_, "cells_combine", 8.0 * ncols * (double)R * (double)M);
    dim3 grid((unsigned)((M + 255) / 256), R, ncols);
    cells_combine_kernel<<<grid, 256, 0, L_.stream>>>(d_w, d_vinv, vinv_pitch, R, ncols, M, d_coef, coef_stride);
}

void pack30(const Launch& L_, const uint32_t* d_felts, size_t n_felts, uint8_t* d_out, size_t len) {
    if (len == 0) return;
    Scope scope(L_, "pack30", 4.0 * (double)n_felts + (double)len);
    size_t n_dw = (len + 3) / 4;
    pack30_kernel<<<(unsigned)((n_dw + 255) / 256), 256, 0, L_.stream>>>(d_felts, n_felts, d_out, len);
}


// ------------------------------------------------------------------------------------------------
// Reconstruction from MANY scattered cells: the R x R system on the device (257 <= R <= 4096)
// ------------------------------------------------------------------------------------------------
// The host inverts the cell matrix V[r][u] = prod_{b in u} s_b(c_r) by Gauss-Jordan for R <= 256 (capi.cpp).  Beyond that the
// cubic cost belongs on the device: blocked Gauss-Jordan on the augmented matrix [V | I] (R x 2R words, row-major), NB pivots
// per round, so that the matrix is streamed through the chip R / NB times instead of R times:
//   panel   one workgroup runs Gauss-Jordan with row pivoting on the R x NB panel (a copy) augmented with the NB x NB identity
//           on the pivot rows; it records the row swaps and ends with Q (R x NB): the inverse D of the pivot block on the pivot
//           rows, -F D on every other row (F = that row's panel entries);
//   swap    the recorded swaps are applied to the full rows and the NB pivot rows are saved;
//   update  every row becomes (itself, unless it is a pivot row) + Q[row] x (saved pivot rows): one rank-NB update of the whole
//           matrix, 64-bit accumulators reduced every third product.
// The left half ends as the identity, the right half as V^-1 (row pitch 2R), which cells_combine consumes directly.
namespace {
constexpr uint32_t GJ_NB = 32;

// V | I for the given cells: row r from cell index c = cell_index[r]; s_b(c) = +- T_{m+b-1}[c >> (b+1)], minus when bit b of c is set
__global__ __launch_bounds__(256) void cells_matrix_kernel(const uint32_t* __restrict__ cell_index, uint32_t R, uint32_t j_bits, uint32_t m,
                                                           uint32_t n, const uint32_t* __restrict__ tw, uint32_t* __restrict__ M) {
    __shared__ uint32_t s[16];
    const uint32_t r = blockIdx.x, c = cell_index[r];
    if (threadIdx.x < j_bits) {
        const uint32_t b = threadIdx.x;
        uint32_t v;
        if (m + b == 0) {  // single points: bit 0 of the index is the circle layer, Y[c >> 1] = [y, -y, -x, x] of a level-0 pair (n >= 3 here)
            const uint32_t h = c >> 1, rr = h & 3u;
            v = tw[2 * (h >> 2) + (rr < 2 ? 1 : 0)];
            if (rr == 1 || rr == 2) v = m31_neg(v);
        } else {
            v = tw[tw_level_offset_dev(n, m + b - 1) + (c >> (b + 1))];
        }
        s[b] = ((c >> b) & 1u) ? m31_neg(v) : v;
    }
    __syncthreads();
    uint32_t* row = M + (size_t)r * 2 * R;
    for (uint32_t u = threadIdx.x; u < R; u += 256) {
        uint32_t prod = 1;
        for (uint32_t b = 0; b < j_bits; b++)
            if ((u >> b) & 1u) prod = m31_mul(prod, s[b]);
        row[u] = prod;
        row[R + u] = u == r ? 1u : 0u;
    }
}

struct GjArgs {
    uint32_t* M;       // R x 2R
    uint32_t* panel;   // R x 2 NB scratch: [panel copy | Q]
    uint32_t* pivrows; // NB x 2R scratch: the saved pivot rows
    uint32_t* state;   // [0] singular flag, [1 .. 1 + NB) the pivot row chosen for each panel column of this round
    uint32_t R, k0;
};

__global__ __launch_bounds__(1024) void gj_panel_kernel(GjArgs a) {
    __shared__ uint32_t s_p;
    __shared__ uint32_t s_prow[2 * GJ_NB];
    __shared__ uint32_t s_f[4096];  // this step's elimination factors (R <= 2^FRIEDA_MAX_LOG_CELLS)
    const uint32_t t = threadIdx.x, R = a.R, W = 2 * GJ_NB;
    if (a.state[0]) return;
    uint32_t* P = a.panel;
    // copy: P[i] = [ M[i][k0 .. k0 + NB) | (i - k0 == j) ]
    for (uint32_t e = t; e < R * W; e += 1024) {
        const uint32_t i = e / W, j = e % W;
        P[e] = j < GJ_NB ? a.M[(size_t)i * 2 * R + a.k0 + j] : ((i - a.k0) == (j - GJ_NB) ? 1u : 0u);
    }
    __syncthreads();
    for (uint32_t j = 0; j < GJ_NB; j++) {
        const uint32_t k = a.k0 + j;
        if (t == 0) s_p = ~0u;
        __syncthreads();
        for (uint32_t i = k + t; i < R; i += 1024)
            if (P[(size_t)i * W + j]) atomicMin(&s_p, i);
        __syncthreads();
        const uint32_t p = s_p;
        if (p == ~0u) {  // no pivot: these cells do not determine the polynomial
            if (t == 0) a.state[0] = 1;
            return;
        }
        if (t == 0) a.state[1 + j] = p;
        // Swap rows k and p, then scale the new row k by 1 / pivot (kept in LDS).  The right half accumulates Q = T' x Iaug, where
        // T' is the elimination that plain Gauss-Jordan would perform on the ALREADY permuted panel (the full matrix gets the
        // swaps first, then T').  A swap found late therefore has to commute past the eliminations done so far:
        // swap . T = (swap T swap) . swap, and conjugating T by the swap moves rows k and p of (T - I) while the identity
        // stays in place — so the right half is swapped with each row's own unit entry taken out before and put back after.
        if (t < W) {
            uint32_t vk = P[(size_t)k * W + t], vp = P[(size_t)p * W + t];
            if (t >= GJ_NB && p != k) {
                const uint32_t jj = t - GJ_NB;
                const uint32_t ik = jj == j ? 1u : 0u, ip = (p - a.k0) == jj ? 1u : 0u;
                const uint32_t zk = m31_sub(vk, ik), zp = m31_sub(vp, ip);  // rows of T - I
                vk = m31_add(zk, ip);  // goes to row p
                vp = m31_add(zp, ik);  // goes to row k
            }
            s_prow[t] = vp;
            if (p != k) P[(size_t)p * W + t] = vk;
        }
        __syncthreads();
        const uint32_t inv = m31_inv(s_prow[j]);
        __syncthreads();
        if (t < W) {
            const uint32_t v = m31_mul(s_prow[t], inv);
            s_prow[t] = v;
            P[(size_t)k * W + t] = v;
        }
        __syncthreads();
        // eliminate column j from every other row.  The factors (column j of the panel, R <= 4096 words) are staged in LDS first,
        // so that no lane reads an entry another lane of the same sweep overwrites; then 32 threads per row (two columns each),
        // 32 rows per sweep
        for (uint32_t i = t; i < R; i += 1024) s_f[i] = i == k ? 0u : P[(size_t)i * W + j];
        __syncthreads();
        const uint32_t lane = t & 31, rsub = t >> 5;
        for (uint32_t i = rsub; i < R; i += 32) {
            const uint32_t f = s_f[i];
            if (f) {
                uint32_t* row = P + (size_t)i * W;
                const uint32_t c0 = lane, c1 = lane + 32;
                row[c0] = m31_sub(row[c0], m31_mul(f, s_prow[c0]));
                row[c1] = m31_sub(row[c1], m31_mul(f, s_prow[c1]));
            }
        }
        __syncthreads();
    }
}

// apply this round's row swaps to the full rows (in pivot order) and save the NB pivot rows
__global__ __launch_bounds__(256) void gj_swap_kernel(GjArgs a) {
    if (a.state[0]) return;
    const uint32_t c = blockIdx.x * 256 + threadIdx.x, W = 2 * a.R;
    if (c >= W) return;
    for (uint32_t j = 0; j < GJ_NB; j++) {
        const uint32_t k = a.k0 + j, p = a.state[1 + j];
        uint32_t* rk = a.M + (size_t)k * W + c;
        if (p != k) {
            uint32_t* rp = a.M + (size_t)p * W + c;
            const uint32_t vk = *rk, vp = *rp;
            *rk = vp;
            *rp = vk;
        }
        a.pivrows[(size_t)j * W + c] = *rk;
    }
}

// M[i][c] = (i is a pivot row ? 0 : M[i][c]) + sum_j Q[i][j] * pivrows[j][c]; tile = 64 rows x 256 columns
__global__ __launch_bounds__(256) void gj_update_kernel(GjArgs a) {
    __shared__ uint32_t sQ[64 * GJ_NB];
    if (a.state[0]) return;
    const uint32_t W = 2 * a.R, c = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * 64;
    for (uint32_t e = threadIdx.x; e < 64 * GJ_NB; e += 256) {
        const uint32_t i = r0 + e / GJ_NB;
        sQ[e] = i < a.R ? a.panel[(size_t)i * 2 * GJ_NB + GJ_NB + e % GJ_NB] : 0u;
    }
    __syncthreads();
    if (c >= W) return;
    uint32_t piv[GJ_NB];
#pragma unroll
    for (uint32_t j = 0; j < GJ_NB; j++) piv[j] = a.pivrows[(size_t)j * W + c];
    for (uint32_t ii = 0; ii < 64 && r0 + ii < a.R; ii++) {
        const uint32_t i = r0 + ii;
        const bool is_piv = i >= a.k0 && i < a.k0 + GJ_NB;
        uint64_t acc = is_piv ? 0u : a.M[(size_t)i * W + c];
#pragma unroll
        for (uint32_t j = 0; j < GJ_NB; j++) {
            acc += (uint64_t)sQ[ii * GJ_NB + j] * piv[j];  // each product < 2^62
            if (j % 3 == 2) acc = m31_reduce64(acc);       // three products + a reduced value stay below 2^64
        }
        a.M[(size_t)i * W + c] = m31_reduce64(acc);
    }
}
}  // namespace

// scratch layout (words): [V | I] R x 2R, panel R x 2 NB, saved pivot rows NB x 2R, state 64, cell indices R
static size_t cells_inverse_index_offset_words(uint32_t R) { return (size_t)R * 2 * R + (size_t)R * 2 * GJ_NB + (size_t)GJ_NB * 2 * R + 64; }
size_t cells_inverse_scratch_bytes(uint32_t R) { return sizeof(uint32_t) * (cells_inverse_index_offset_words(R) + R) + 256; }
// where the caller uploads the R cell indices (inside the scratch block)
uint32_t* cells_inverse_index_buffer(uint8_t* d_scratch, uint32_t R) {
    return reinterpret_cast<uint32_t*>(d_scratch) + cells_inverse_index_offset_words(R);
}

// d_scratch: cells_inverse_scratch_bytes(R) bytes, 256-byte aligned.  On return (asynchronously) the inverse sits at
// *d_vinv_out with row pitch *pitch_out words; d_state_out[0] != 0 reports a singular system.  R a multiple of GJ_NB.
void cells_matrix_inverse_device(const Launch& L_, const uint32_t* d_cell_index, uint32_t R, uint32_t j_bits, uint32_t m, uint32_t n,
                                 const uint32_t* d_tw, uint8_t* d_scratch, const uint32_t** d_vinv_out, size_t* pitch_out,
                                 const uint32_t** d_state_out) {
    GjArgs a{};
    a.M = reinterpret_cast<uint32_t*>(d_scratch);
    a.panel = a.M + (size_t)R * 2 * R;
    a.pivrows = a.panel + (size_t)R * 2 * GJ_NB;
    a.state = a.pivrows + (size_t)GJ_NB * 2 * R;
    a.R = R;
    hipStream_t s = L_.stream;
    Scope scope(L_, "cells_inverse", 8.0 * (double)R * 2.0 * (double)R * (double)(R / GJ_NB));
    (void)hipMemsetAsync(a.state, 0, 64 * sizeof(uint32_t), s);
    cells_matrix_kernel<<<R, 256, 0, s>>>(d_cell_index, R, j_bits, m, n, d_tw, a.M);
    for (uint32_t k0 = 0; k0 < R; k0 += GJ_NB) {
        a.k0 = k0;
        gj_panel_kernel<<<1, 1024, 0, s>>>(a);
        gj_swap_kernel<<<(2 * R + 255) / 256, 256, 0, s>>>(a);
        gj_update_kernel<<<dim3((2 * R + 255) / 256, (R + 63) / 64), 256, 0, s>>>(a);
    }
    *d_vinv_out = a.M + R;
    *pitch_out = 2 * (size_t)R;
    *d_state_out = a.state;
}

}  // namespace k
}  // namespace frieda
