// fri.hip — FRI fold kernels, decommitment gather and proof-of-work grind (gfx950).
//
// Replaces, behind `FriProver::commit` / `decommit` and `CpuBackend::grind`
// (/root/reference/src/proof.rs:52-66):
//   * stwo backend/cpu/fri.rs::fold_circle_into_line and ::fold_line — per adjacent pair (2i, 2i+1) of the
//     bit-reversed evaluation: (f0, f1) = (a + b, (a - b) * itw), out = f0 + alpha * f1 (the circle variant
//     accumulates dst * alpha^2 + out).  The CPU backend recomputes the domain point (a 31-step scalar
//     multiplication) and a field inversion per pair; here the inverse twiddle is one table read:
//     circle pairs use inverse-Y[i] = [iy, -iy, -ix, ix] from level 0 of the inverse table, line layers of
//     log size m use inverse level n-1-m directly.
//   * MerkleProver::decommit's random reads — one gather launch for all layers' witness words and hashes.
//   * stwo backend/cpu/grind.rs — the sequential nonce scan becomes a chunked parallel scan with atomicMin,
//     so the *smallest* qualifying nonce is returned, as the reference does.
// Columns are SoA (4 M31 coordinate columns of a QM31 column), so a thread's pair is one 8-byte load per
// coordinate and every access is coalesced.  HBM-bound: 32 B read + 16 B written per output (+4 B twiddle).
#include <hip/hip_runtime.h>

#include "blake2s.h"
#include "kernels.h"

namespace frieda {
namespace k {

namespace {

constexpr int FR_THREADS = 256;

__device__ __forceinline__ uint32_t inv_circle_twiddle(const uint32_t* __restrict__ itw, uint32_t n, size_t i, uint32_t inv_init_y) {
    if (n < 3) return (i & 1u) ? m31_neg(inv_init_y) : inv_init_y;
    size_t j = i >> 2;
    uint32_t r = (uint32_t)(i & 3u);
    uint32_t v = itw[2 * j + (r < 2 ? 1 : 0)];
    return (r == 1 || r == 2) ? m31_neg(v) : v;
}

__device__ __forceinline__ void load_pair(const uint32_t* __restrict__ src, size_t stride, size_t i, QM31& f0, QM31& f1) {
    uint2 a = reinterpret_cast<const uint2*>(src)[i];
    uint2 b = reinterpret_cast<const uint2*>(src + stride)[i];
    uint2 c = reinterpret_cast<const uint2*>(src + 2 * stride)[i];
    uint2 d = reinterpret_cast<const uint2*>(src + 3 * stride)[i];
    f0 = {a.x, b.x, c.x, d.x};
    f1 = {a.y, b.y, c.y, d.y};
}

__global__ __launch_bounds__(FR_THREADS) void fold_circle_kernel(uint32_t* __restrict__ dst, size_t dst_stride,
                                                                 const uint32_t* __restrict__ src, size_t src_stride,
                                                                 uint32_t n, const uint32_t* __restrict__ itw,
                                                                 uint32_t inv_init_y, QM31 alpha, QM31 alpha_sq) {
    size_t i = (size_t)blockIdx.x * FR_THREADS + threadIdx.x;
    size_t half = (size_t)1 << (n - 1);
    if (i >= half) return;
    QM31 a, b;
    load_pair(src, src_stride, i, a, b);
    uint32_t it = inv_circle_twiddle(itw, n, i, inv_init_y);
    QM31 f0 = qm_add(a, b), f1 = qm_scale(qm_sub(a, b), it);
    QM31 fp = qm_add(qm_mul(alpha, f1), f0);
    QM31 d = {dst[i], dst[dst_stride + i], dst[2 * dst_stride + i], dst[3 * dst_stride + i]};
    QM31 r = qm_add(qm_mul(d, alpha_sq), fp);
    dst[i] = r.a;
    dst[dst_stride + i] = r.b;
    dst[2 * dst_stride + i] = r.c;
    dst[3 * dst_stride + i] = r.d;
}

__global__ __launch_bounds__(FR_THREADS) void fold_line_kernel(const uint32_t* __restrict__ src, size_t src_stride,
                                                               uint32_t m, const uint32_t* __restrict__ itw_level,
                                                               QM31 alpha, uint32_t* __restrict__ dst, size_t dst_stride) {
    size_t i = (size_t)blockIdx.x * FR_THREADS + threadIdx.x;
    size_t half = (size_t)1 << (m - 1);
    if (i >= half) return;
    QM31 a, b;
    load_pair(src, src_stride, i, a, b);
    uint32_t it = itw_level[i];
    QM31 f0 = qm_add(a, b), f1 = qm_scale(qm_sub(a, b), it);
    QM31 r = qm_add(f0, qm_mul(alpha, f1));
    dst[i] = r.a;
    dst[dst_stride + i] = r.b;
    dst[2 * dst_stride + i] = r.c;
    dst[3 * dst_stride + i] = r.d;
}

__global__ __launch_bounds__(FR_THREADS) void gather_kernel(const uint32_t* __restrict__ base,
                                                            const uint64_t* __restrict__ word_idx, size_t n_words,
                                                            uint32_t* __restrict__ out_words,
                                                            const uint64_t* __restrict__ hash_idx, size_t n_hashes,
                                                            uint8_t* __restrict__ out_hashes) {
    size_t t = (size_t)blockIdx.x * FR_THREADS + threadIdx.x;
    if (t < n_words) out_words[t] = base[word_idx[t]];
    // one thread per 16-byte half of a hash
    if (t < 2 * n_hashes) {
        size_t hsh = t >> 1, part = t & 1;
        const uint4* srcp = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(base) + 32 * hash_idx[hsh]);
        reinterpret_cast<uint4*>(out_hashes)[2 * hsh + part] = srcp[part];
    }
}

struct Digest8 {
    uint32_t w[8];
};

__global__ __launch_bounds__(FR_THREADS) void grind_kernel(Digest8 dg, uint32_t pow_bits, unsigned long long base,
                                                           unsigned long long count, unsigned long long* result) {
    unsigned long long t = (unsigned long long)blockIdx.x * FR_THREADS + threadIdx.x;
    if (t >= count) return;
    unsigned long long nonce = base + t;
    // Blake2sChannel::mix_u64: compress(h = digest, m = [lo, hi, 0...], t = f = 0); trailing_zeros of the low 128 bits
    uint32_t m[16] = {(uint32_t)nonce, (uint32_t)(nonce >> 32), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t h[8], r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = dg.w[i];
    b2_compress_tp<FRIEDA_B2_IDLE_GRIND>(h, m, 0, 0, 0, 0, r);  // (a chip-filling launch: the throughput form, blake2s.h)
    uint32_t tz;
    if (r[0])
        tz = __ffs(r[0]) - 1;
    else if (r[1])
        tz = 32 + __ffs(r[1]) - 1;
    else if (r[2])
        tz = 64 + __ffs(r[2]) - 1;
    else if (r[3])
        tz = 96 + __ffs(r[3]) - 1;
    else
        tz = 128;
    if (tz >= pow_bits) atomicMin(result, nonce);
}

}  // namespace

void fold_circle_into_line(const Launch& L, uint32_t* d_dst, size_t dst_stride, const uint32_t* d_src, size_t src_stride,
                           uint32_t n, const uint32_t* d_itw, DomainScalars ds, Alpha alpha) {
    QM31 a = {alpha.v[0], alpha.v[1], alpha.v[2], alpha.v[3]};
    QM31 a2 = qm_mul(a, a);
    size_t half = (size_t)1 << (n - 1);
    hipStream_t s = L.stream;
    Scope scope(L, "fold_circle", 48.0 * (double)half);  // 16N in + 8N out (24N)
    fold_circle_kernel<<<(unsigned)((half + FR_THREADS - 1) / FR_THREADS), FR_THREADS, 0, s>>>(d_dst, dst_stride, d_src, src_stride,
                                                                                             n, d_itw, ds.inv_init_y, a, a2);
}

void fold_line(const Launch& L, const uint32_t* d_src, size_t src_stride, uint32_t m, uint32_t n, const uint32_t* d_itw,
               DomainScalars ds, Alpha alpha, uint32_t* d_dst, size_t dst_stride) {
    (void)ds;
    QM31 a = {alpha.v[0], alpha.v[1], alpha.v[2], alpha.v[3]};
    size_t half = (size_t)1 << (m - 1);
    // line domain of log size m = coset half_odds(n-1) doubled n-1-m times = twiddle level n-1-m (size 2^(m-1))
    const uint32_t* lvl = d_itw + tw_level_offset(n, n - 1 - m);
    hipStream_t s = L.stream;
    Scope scope(L, "fold_line", 48.0 * (double)half);  // 16M in + 8M out
    fold_line_kernel<<<(unsigned)((half + FR_THREADS - 1) / FR_THREADS), FR_THREADS, 0, s>>>(d_src, src_stride, m, lvl, a, d_dst,
                                                                                           dst_stride);
}

void gather(const Launch& L, const uint32_t* d_base, const uint64_t* d_word_idx, size_t n_words, uint32_t* d_out_words,
            const uint64_t* d_hash_idx, size_t n_hashes, uint8_t* d_out_hashes) {
    size_t threads = n_words > 2 * n_hashes ? n_words : 2 * n_hashes;
    if (threads == 0) return;
    hipStream_t s = L.stream;
    Scope scope(L, "gather", 16.0 * (double)n_words + 72.0 * (double)n_hashes);
    gather_kernel<<<(unsigned)((threads + FR_THREADS - 1) / FR_THREADS), FR_THREADS, 0, s>>>(d_base, d_word_idx, n_words,
                                                                                           d_out_words, d_hash_idx, n_hashes,
                                                                                           d_out_hashes);
}

void grind_scan(const Launch& L, const uint32_t digest[8], uint32_t pow_bits, uint64_t base, uint64_t count,
                unsigned long long* d_result) {
    Digest8 dg;
    for (int i = 0; i < 8; i++) dg.w[i] = digest[i];
    hipStream_t s = L.stream;
    Scope scope(L, "grind", 0.0);  // compute only
    grind_kernel<<<(unsigned)((count + FR_THREADS - 1) / FR_THREADS), FR_THREADS, 0, s>>>(dg, pow_bits, base, count, d_result);
}

}  // namespace k
}  // namespace frieda
