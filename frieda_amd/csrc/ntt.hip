// ntt.hip — Reed–Solomon encode: radix-2 circle FFT over M31 (gfx950).
//
// Replaces `SecureCirclePoly::evaluate_with_twiddles` -> 4 x `CpuBackend::evaluate`
// (/root/reference/src/commit.rs:16, src/proof.rs:48-49; stwo backend/cpu/circle.rs::evaluate,
// core/fft.rs::butterfly).  Input: 2^L coefficients per column in natural order (zero-extended to 2^n by
// the reference); output: 2^n evaluations per column in bit-reversed domain order.
//
// Structure.  Layer i pairs indices that differ in bit i; the twiddle of a pair depends only on the index
// bits above i.  Layers run i = n-1 .. 1 (line layers, twiddle T_{i-1}[idx >> (i+1)]) then i = 0 (circle
// layer, twiddle Y[idx >> 1]).  Because the coefficient vector is zero above 2^L, the top n-L layers are
// butterflies against zero: they only replicate the coefficient block 2^(n-L) times.  They are never
// executed — the first real pass reads coefficient `idx mod 2^L` instead.
//
// Passes.  A pass executes a run of layers i_hi..i_lo on tiles held in LDS: the tile is the 2^t (t =
// i_hi-i_lo+1) values of bits [i_lo, i_hi] times 2^log_w consecutive values of the low bits (so that global
// accesses are 2^log_w-word contiguous runs), all other index bits fixed per workgroup.  The last pass has
// i_lo = 0 and reads/writes fully contiguous tiles.  HBM traffic per pass: 4 B read + 4 B written per
// element; all 4 columns share the twiddle tables (grid.y = column).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace frieda {
namespace k {

namespace {

constexpr int NTT_THREADS = 256;

// circle-layer twiddle Y[h] from the first line level: pairs (x, y) -> [y, -y, -x, x]
__device__ __forceinline__ uint32_t circle_twiddle(const uint32_t* __restrict__ tw, uint32_t n, uint32_t h, uint32_t init_y) {
    if (n < 3) return (h & 1u) ? m31_neg(init_y) : init_y;  // n == 1: [y]; n == 2: [y, -y]
    uint32_t j = h >> 2, r = h & 3u;
    uint32_t v = tw[2 * j + (r < 2 ? 1 : 0)];
    return (r == 1 || r == 2) ? m31_neg(v) : v;
}

__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(const uint32_t* __restrict__ in, size_t in_stride,
                                                               uint32_t in_mask, uint32_t* __restrict__ out,
                                                               size_t out_stride, const uint32_t* __restrict__ tw,
                                                               uint32_t n, uint32_t i_hi, uint32_t i_lo, uint32_t log_w,
                                                               uint32_t init_y) {
    extern __shared__ uint32_t lds[];
    const uint32_t t = i_hi - i_lo + 1;
    const uint32_t tile_log = t + log_w;
    const uint32_t tile = 1u << tile_log;
    const uint32_t wmask = (1u << log_w) - 1;
    const uint32_t nwb_log = i_lo - log_w;  // number of w-blocks per (hblk) = 2^(i_lo - log_w)
    const uint32_t wblk = blockIdx.x & ((1u << nwb_log) - 1);
    const uint32_t hblk = blockIdx.x >> nwb_log;
    const uint32_t gbase = (hblk << (i_hi + 1)) | (wblk << log_w);
    in += (size_t)blockIdx.y * in_stride;
    out += (size_t)blockIdx.y * out_stride;

    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t g = gbase | ((e >> log_w) << i_lo) | (e & wmask);
        lds[e] = in[g & in_mask];
    }
    __syncthreads();

    for (int i = (int)i_hi; i >= (int)i_lo; i--) {
        const uint32_t s = (uint32_t)i - i_lo;
        const uint32_t* lvl = (i >= 1) ? tw + tw_level_offset_dev(n, (uint32_t)i - 1) : tw;
        for (uint32_t b = threadIdx.x; b < (tile >> 1); b += NTT_THREADS) {
            uint32_t w = b & wmask, bj = b >> log_w;
            uint32_t j0 = ((bj >> s) << (s + 1)) | (bj & ((1u << s) - 1));
            uint32_t h = (hblk << (i_hi - (uint32_t)i)) | (j0 >> (s + 1));
            uint32_t twv = (i >= 1) ? lvl[h] : circle_twiddle(tw, n, h, init_y);
            uint32_t e0 = (j0 << log_w) | w, e1 = e0 + (1u << (s + log_w));
            uint32_t a = lds[e0], tt = m31_mul(lds[e1], twv);
            lds[e0] = m31_add(a, tt);
            lds[e1] = m31_sub(a, tt);
        }
        __syncthreads();
    }

    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t g = gbase | ((e >> log_w) << i_lo) | (e & wmask);
        out[g] = lds[e];
    }
}

// pure replication (L == 0: a constant polynomial has no real layers)
__global__ void ntt_broadcast_kernel(const uint32_t* __restrict__ in, size_t in_stride, uint32_t* __restrict__ out,
                                     size_t out_stride, size_t n_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) out[(size_t)blockIdx.y * out_stride + i] = in[(size_t)blockIdx.y * in_stride];
}

constexpr uint32_t LAST_PASS_MAX_LOG = 12;  // 16 KiB tile
constexpr uint32_t MID_PASS_MAX_LOG = 8;    // 2^8 x 16 words = 16 KiB tile
constexpr uint32_t MID_LOG_W = 4;           // 64-byte contiguous runs

}  // namespace

void circle_evaluate(const Launch& L_, const uint32_t* d_coef, size_t coef_stride, uint32_t ncols, uint32_t L, uint32_t n,
                     const uint32_t* d_tw, DomainScalars ds, uint32_t* d_out, size_t out_stride) {
    const size_t N = (size_t)1 << n;
    hipStream_t s = L_.stream;
    // algorithmic bytes of the encode: read 2^L, write 2^n words per column (16N(1 + 2^-B) for 4 columns),
    // attributed to the passes in proportion to the elements they move
    const double enc_bytes = 4.0 * ncols * ((double)N + (double)((size_t)1 << L));
    if (L == 0) {
        Scope scope(L_, "ntt_broadcast", enc_bytes);
        dim3 grid((unsigned)((N + 255) / 256), ncols);
        ntt_broadcast_kernel<<<grid, 256, 0, s>>>(d_coef, coef_stride, d_out, out_stride, N);
        return;
    }
    // real layers i = L-1 .. 0; the last pass takes up to LAST_PASS_MAX_LOG of them, the earlier passes split the rest
    uint32_t last_t = L < LAST_PASS_MAX_LOG ? L : LAST_PASS_MAX_LOG;
    uint32_t rest = L - last_t;
    uint32_t n_mid = (rest + MID_PASS_MAX_LOG - 1) / MID_PASS_MAX_LOG;
    const uint32_t* src = d_coef;
    size_t src_stride = coef_stride;
    uint32_t src_mask = (uint32_t)(((size_t)1 << L) - 1);
    uint32_t i_hi = L - 1;
    for (uint32_t p = 0; p < n_mid; p++) {
        uint32_t t = (rest + (n_mid - p) - 1) / (n_mid - p);  // even split of what is left
        uint32_t i_lo = i_hi + 1 - t;
        uint32_t log_w = MID_LOG_W;  // i_lo >= last_t >= MID_LOG_W whenever a mid pass exists (L > 12)
        dim3 grid((unsigned)(N >> (t + log_w)), ncols);
        size_t lds_bytes = (size_t)4 << (t + log_w);
        Scope scope(L_, "ntt_pass_mid", enc_bytes / (n_mid + 1));
        ntt_pass_kernel<<<grid, NTT_THREADS, lds_bytes, s>>>(src, src_stride, src_mask, d_out, out_stride, d_tw, n, i_hi, i_lo,
                                                            log_w, ds.init_y);
        src = d_out;
        src_stride = out_stride;
        src_mask = (uint32_t)(N - 1);
        rest -= t;
        i_hi = i_lo - 1;
    }
    {
        uint32_t t = last_t;  // i_hi == t - 1, i_lo == 0
        dim3 grid((unsigned)(N >> t), ncols);
        size_t lds_bytes = (size_t)4 << t;
        Scope scope(L_, "ntt_pass_last", enc_bytes / (n_mid + 1));
        ntt_pass_kernel<<<grid, NTT_THREADS, lds_bytes, s>>>(src, src_stride, src_mask, d_out, out_stride, d_tw, n, i_hi, 0, 0,
                                                            ds.init_y);
    }
}

}  // namespace k
}  // namespace frieda
