// polyops.hip — the trait methods of the plug-in surface that frieda's three functions never call (gfx950).
//
// SURVEY.md §8(b) lists stwo's backend traits behind `CpuBackend` (/root/reference/src/commit.rs:15-17, src/proof.rs:47-58) as
// PolyOps{precompute_twiddles, evaluate, interpolate, extend, eval_at_point}, FriOps{fold_line, fold_circle_into_line, decompose}.
// frieda's path calls the first three and the two folds (twiddle.hip, ntt.hip, intt.hip, fri.hip); an `impl PolyOps / FriOps for
// HipBackend` also needs the other three, or `unimplemented!()` bodies.  They are here, restated from
// stwo-prover@19d12d7's published CpuBackend (backend/cpu/circle.rs, backend/cpu/fri.rs, core/poly/utils.rs::fold):
//   * extend(poly, log_size): the coefficient vector zero-extended to 2^log_size.
//   * eval_at_point(poly, point): sum_j coeff[j] * prod_{b in bits(j)} F_b with F_0 = point.y, F_1 = point.x, F_{b+1} = 2 F_b^2 - 1
//     over QM31 (stwo folds the halves recursively with the factors reversed; field arithmetic is exact, so the order of the
//     additions is free).  A two-level reduction: a workgroup folds a tile of 4096 inputs (16 per thread in registers, strided so
//     the loads coalesce, then 8 tree steps through LDS), the tiles' results are folded by the same kernel again.
//   * decompose(eval) -> (g, lambda): lambda = (sum over the first half - sum over the second half of the bit-reversed evaluation) /
//     domain size; g = eval - lambda on the first half, eval + lambda on the second.
// None of this is on the commit / prove hot path; the kernels are HBM-bound one-pass reductions.
#include <hip/hip_runtime.h>

#include "field.h"
#include "kernels.h"

namespace frieda {
namespace k {

namespace {
constexpr int PO_THREADS = 256;
constexpr uint32_t EV_TILE_LOG = 12;  // inputs per workgroup of the fold kernel

__global__ __launch_bounds__(PO_THREADS) void extend_kernel(const uint32_t* __restrict__ in, size_t n_in, uint32_t* __restrict__ out, size_t n_out) {
    const size_t i = (size_t)blockIdx.x * PO_THREADS + threadIdx.x;
    const size_t col = blockIdx.y;
    if (i < n_out) out[col * n_out + i] = i < n_in ? in[col * n_in + i] : 0u;
}

// One level of the two-level reduction.  Input of column `col`: SECURE_IN ? in[col][4][n_in] (SoA QM31) : in[col][n_in] (M31).
// Workgroup w folds inputs w * 2^tile_log .. (w + 1) * 2^tile_log with the factors f.f[bit0 ..]; out[col][4][n_in >> tile_log].
template <bool SECURE_IN>
__global__ __launch_bounds__(PO_THREADS) void eval_fold_kernel(const uint32_t* __restrict__ in, size_t n_in, uint32_t tile_log, uint32_t bit0, EvalFactors f,
                                                               uint32_t* __restrict__ out) {
    __shared__ uint32_t sh[4][PO_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t col = blockIdx.y, w = blockIdx.x;
    const size_t n_out = n_in >> tile_log;
    const uint32_t lane_log = tile_log < 8 ? tile_log : 8;  // bits of the in-tile index that are the thread index
    const uint32_t reg_log = tile_log - lane_log;           // bits folded in registers
    const uint32_t* src = in + col * (SECURE_IN ? 4 : 1) * n_in + (w << tile_log);
    QM31 v[1u << (EV_TILE_LOG - 8)];
    const bool live = t < (1u << lane_log);
#pragma unroll
    for (uint32_t e = 0; e < (1u << (EV_TILE_LOG - 8)); e++) {
        v[e] = QM31{0, 0, 0, 0};
        if (live && e < (1u << reg_log)) {
            const size_t i = t + ((size_t)e << lane_log);
            if (SECURE_IN)
                v[e] = QM31{src[i], src[n_in + i], src[2 * n_in + i], src[3 * n_in + i]};
            else
                v[e] = QM31{src[i], 0, 0, 0};
        }
    }
    // register levels: in-tile bit lane_log + b pairs v[2 i] and v[2 i + 1]
#pragma unroll
    for (uint32_t b = 0; b < EV_TILE_LOG - 8; b++) {
        if (b < reg_log) {
            const QM31 fb = f.f[bit0 + lane_log + b];
#pragma unroll
            for (uint32_t i = 0; i < (1u << (EV_TILE_LOG - 8 - 1 - b)); i++) v[i] = qm_add(v[2 * i], qm_mul(v[2 * i + 1], fb));
        }
    }
    sh[0][t] = v[0].a, sh[1][t] = v[0].b, sh[2][t] = v[0].c, sh[3][t] = v[0].d;
    __syncthreads();
    // lane levels: in-tile bit s pairs thread t (bit s clear) with thread t + 2^s
    for (uint32_t s = 0; s < lane_log; s++) {
        const uint32_t span = 2u << s;
        if (live && (t & (span - 1)) == 0) {
            const uint32_t o = t + (1u << s);
            const QM31 hi{sh[0][o], sh[1][o], sh[2][o], sh[3][o]};
            const QM31 lo{sh[0][t], sh[1][t], sh[2][t], sh[3][t]};
            const QM31 r = qm_add(lo, qm_mul(hi, f.f[bit0 + s]));
            sh[0][t] = r.a, sh[1][t] = r.b, sh[2][t] = r.c, sh[3][t] = r.d;
        }
        __syncthreads();
    }
    if (t == 0) {
        uint32_t* o = out + col * 4 * n_out + w;
        o[0] = sh[0][0], o[n_out] = sh[1][0], o[2 * n_out] = sh[2][0], o[3 * n_out] = sh[3][0];
    }
}

// signed sums of one coordinate column: first half counted +, second half -; partial[coord][wg] = {plus, minus} mod P
__global__ __launch_bounds__(PO_THREADS) void decompose_sum_kernel(const uint32_t* __restrict__ ev, size_t n, uint32_t per_thread, uint32_t* __restrict__ partial) {
    __shared__ unsigned long long shp[PO_THREADS], shm[PO_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t coord = blockIdx.y, half = n / 2;
    const uint32_t* col = ev + coord * n;
    const size_t base = (size_t)blockIdx.x * PO_THREADS * per_thread;
    unsigned long long plus = 0, minus = 0;  // per_thread <= 2^16 values below 2^31: no overflow
    for (uint32_t e = 0; e < per_thread; e++) {
        const size_t i = base + (size_t)e * PO_THREADS + t;
        if (i < n) {
            if (i < half)
                plus += col[i];
            else
                minus += col[i];
        }
    }
    shp[t] = m31_reduce64(plus), shm[t] = m31_reduce64(minus);
    __syncthreads();
    for (uint32_t s = PO_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) shp[t] += shp[t + s], shm[t] += shm[t + s];  // <= 256 * 2^31
        __syncthreads();
    }
    if (t == 0) {
        uint32_t* o = partial + 2 * (coord * gridDim.x + blockIdx.x);
        o[0] = m31_reduce64(shp[0]), o[1] = m31_reduce64(shm[0]);
    }
}
// one workgroup: lambda[coord] = (sum plus - sum minus) * inv_n
__global__ __launch_bounds__(PO_THREADS) void decompose_lambda_kernel(const uint32_t* __restrict__ partial, uint32_t n_wg, uint32_t inv_n, uint32_t* __restrict__ lambda) {
    __shared__ unsigned long long shp[PO_THREADS], shm[PO_THREADS];
    const uint32_t t = threadIdx.x;
    for (uint32_t coord = 0; coord < 4; coord++) {
        unsigned long long plus = 0, minus = 0;
        for (uint32_t w = t; w < n_wg; w += PO_THREADS) {  // n_wg <= 2^20 values below 2^31 each
            plus += partial[2 * (coord * n_wg + w)];
            minus += partial[2 * (coord * n_wg + w) + 1];
        }
        shp[t] = m31_reduce64(plus), shm[t] = m31_reduce64(minus);
        __syncthreads();
        for (uint32_t s = PO_THREADS / 2; s > 0; s >>= 1) {
            if (t < s) shp[t] += shp[t + s], shm[t] += shm[t + s];
            __syncthreads();
        }
        if (t == 0) lambda[coord] = m31_mul(m31_sub(m31_reduce64(shp[0]), m31_reduce64(shm[0])), inv_n);
        __syncthreads();
    }
}
__global__ __launch_bounds__(PO_THREADS) void decompose_apply_kernel(const uint32_t* __restrict__ ev, size_t n, const uint32_t* __restrict__ lambda,
                                                                     uint32_t* __restrict__ g) {
    const size_t i = (size_t)blockIdx.x * PO_THREADS + threadIdx.x;
    const size_t coord = blockIdx.y;
    if (i >= n) return;
    const uint32_t l = lambda[coord], v = ev[coord * n + i];
    g[coord * n + i] = i < n / 2 ? m31_sub(v, l) : m31_add(v, l);
}
}  // namespace

void circle_extend(const Launch& L, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, uint32_t log_size, uint32_t* d_out) {
    Scope sc(L, "circle_extend", 4.0 * ncols * (((size_t)1 << log_coef) + ((size_t)1 << log_size)));
    const size_t n_out = (size_t)1 << log_size;
    dim3 grid((unsigned)((n_out + PO_THREADS - 1) / PO_THREADS), ncols);
    hipLaunchKernelGGL(extend_kernel, grid, dim3(PO_THREADS), 0, L.stream, d_coef, (size_t)1 << log_coef, d_out, n_out);
}

// two ping-pong buffers of QM31 partials: after the first level 2^(log_coef - 12) per column, after the second 2^(log_coef - 24)
size_t eval_at_point_scratch_bytes(uint32_t ncols, uint32_t log_coef) {
    const size_t n1 = (size_t)1 << (log_coef > EV_TILE_LOG ? log_coef - EV_TILE_LOG : 0);
    const size_t n2 = (size_t)1 << (log_coef > 2 * EV_TILE_LOG ? log_coef - 2 * EV_TILE_LOG : 0);
    return 16 * (size_t)ncols * (n1 + n2) + 512;
}
// d_result: the buffer (inside d_scratch) whose first ncols * 4 words hold the values, column-major [col][4]
const uint32_t* circle_eval_at_point(const Launch& L, const uint32_t* d_coef, uint32_t ncols, uint32_t log_coef, const EvalFactors& f, uint32_t* d_scratch) {
    Scope sc(L, "circle_eval_at_point", 4.0 * ncols * ((size_t)1 << log_coef));
    const size_t n1 = (size_t)1 << (log_coef > EV_TILE_LOG ? log_coef - EV_TILE_LOG : 0);
    uint32_t* buf[2] = {d_scratch, d_scratch + 4 * (size_t)ncols * n1 + 64};
    const uint32_t* in = d_coef;
    uint32_t m = log_coef, bit0 = 0;
    int which = 0;
    bool first = true;
    do {
        const uint32_t tile_log = m < EV_TILE_LOG ? m : EV_TILE_LOG;
        dim3 grid((unsigned)((size_t)1 << (m - tile_log)), ncols);
        if (first)
            hipLaunchKernelGGL(eval_fold_kernel<false>, grid, dim3(PO_THREADS), 0, L.stream, in, (size_t)1 << m, tile_log, bit0, f, buf[which]);
        else
            hipLaunchKernelGGL(eval_fold_kernel<true>, grid, dim3(PO_THREADS), 0, L.stream, in, (size_t)1 << m, tile_log, bit0, f, buf[which]);
        in = buf[which];
        which ^= 1;
        m -= tile_log;
        bit0 += tile_log;
        first = false;
    } while (m > 0);
    return in;
}

size_t decompose_scratch_bytes(uint32_t log_size) {
    const size_t n = (size_t)1 << log_size;
    const size_t per_wg = (size_t)PO_THREADS * 16;
    return 8 * 4 * ((n + per_wg - 1) / per_wg) + 64;
}
// d_scratch: decompose_scratch_bytes; lambda is left in its first four words
void fri_decompose(const Launch& L, const uint32_t* d_eval, uint32_t log_size, uint32_t* d_g, uint32_t* d_scratch) {
    Scope sc(L, "fri_decompose", 4.0 * 4 * 3 * ((size_t)1 << log_size));
    const size_t n = (size_t)1 << log_size;
    const uint32_t per_thread = 16;
    const size_t per_wg = (size_t)PO_THREADS * per_thread;
    const uint32_t n_wg = (uint32_t)((n + per_wg - 1) / per_wg);
    uint32_t* d_lambda = d_scratch;
    uint32_t* d_partial = d_scratch + 16;
    hipLaunchKernelGGL(decompose_sum_kernel, dim3(n_wg, 4), dim3(PO_THREADS), 0, L.stream, d_eval, n, per_thread, d_partial);
    const uint32_t inv_n = m31_inv((uint32_t)(n % P31));
    hipLaunchKernelGGL(decompose_lambda_kernel, dim3(1), dim3(PO_THREADS), 0, L.stream, d_partial, n_wg, inv_n, d_lambda);
    hipLaunchKernelGGL(decompose_apply_kernel, dim3((unsigned)((n + PO_THREADS - 1) / PO_THREADS), 4), dim3(PO_THREADS), 0, L.stream, d_eval, n, d_lambda, d_g);
}

}  // namespace k
}  // namespace frieda
