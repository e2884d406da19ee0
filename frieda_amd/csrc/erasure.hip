// erasure.hip — reconstruction from ANY sufficiently large set of sampled points of the codeword (the README's
// sample() flow, /root/reference/README.md:56-69 — not in /root/reference/src, so parity is against the oracle's restatement and
// encode -> erase -> reconstruct round trips).
//
// intt.hip rebuilds a polynomial from exactly 2^L values by inverting a dense system: cubic in the number of cells, capped at 4096.
// Past that this file uses the erasure-locator route of Reed-Solomon decoding, carried over to the circle domain:
//
//   p         the polynomial: K = 2^L coefficients per column in the circle-FFT basis, degree <= K / 2
//   D         the circle domain of the codeword (canonic coset, N = 2^n points); V_D(x, y) = pi^(n-1)(x), pi(x) = 2 x^2 - 1, vanishes
//             exactly on D (n - 1 doublings take a point of D to x = 0)
//   S         K + 2 of the sampled points (more may be offered: the rest only serve the consistency check at the end)
//   Z_S       the product of the K / 2 + 1 lines through consecutive pairs of S (a line meets the circle in exactly two points):
//             vanishes exactly on S, degree K / 2 + 1
//   Z         = V_D / Z_S: a polynomial function on the circle (the zeros of Z_S are simple zeros of V_D) of degree N / 2 - K / 2 - 1
//             that vanishes exactly on D \ S — the erasure locator — without ever touching the N - K - 2 missing points
//   Z * p     has degree <= N / 2 - 1, so it lies in the space of the size-N circle FFT, and its values on ALL of D are known:
//             Z(s) p(s) on S, zero elsewhere
//
//   1. Z on S: V_D and Z_S both vanish there, so Z(s) is the ratio of their derivatives along the circle's tangent (-y, x):
//             V_D' = -y prod_{j < n-1} 4 pi^j(x);  Z_S' = (-y A + x B) of the point's own line times the other lines' values
//   2. w = Z * p on D (zero off S); inverse circle FFT of size N -> the N coefficients of Z * p
//   3. evaluate Z * p on the next canonic domain D' (2N points, disjoint from D) and take its first block of K entries; there
//      p = (Z p) Z_S / V_D pointwise: K values of p on a sub-coset of D'
//   4. inverse transform of that block (intt.hip) -> the coefficients of p
//   5. encode p again and compare EVERY offered sample (every distinct cell: a repeated index is dropped by the de-duplication, first
//      occurrence wins, and is not compared): samples that are not values of one polynomial are reported, not returned
//
// Cost: (K + 2)(K / 2 + 1) + K (K / 2 + 1) line evaluations for the two products — quadratic in the polynomial, not in the domain —
// plus five transforms.  Exact arithmetic: consistent samples give the polynomial.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace frieda {
namespace k {

namespace {

// point of the circle domain of log size g.n at bit-reversed position `pos` (stwo CircleDomain::at over Coset::half_odds(n - 1):
// natural index j < N / 2 is half_coset[j] = initial + j * step, j >= N / 2 its conjugate)
__device__ __forceinline__ CPoint domain_point(const ErasureDomain& g, uint32_t pos) {
    const uint32_t j = g.n ? (__brev(pos) >> (32 - g.n)) : 0u;
    const bool conj = g.n && ((j >> (g.n - 1)) & 1u);
    const uint32_t jj = g.n ? (j & ((1u << (g.n - 1)) - 1u)) : 0u;
    CPoint p = g.init;
    for (uint32_t b = 0; b + 1 < g.n; b++)
        if ((jj >> b) & 1u) p = cp_add(p, g.step_pow[b]);
    if (conj) p.y = m31_neg(p.y);
    return p;
}

// ---- the caller's cell list -> de-duplicated position lists, on the device ----
// The caller names its samples by cell index (host array, any order, repeats allowed: the first occurrence of a cell counts).  Building
// the de-duplicated (position, source offset) lists on the host cost more than the locator for 2^20 single points (a random-access
// bitmap and 8 MB of fresh vectors per call), so: owner[cell] = atomicMin(index in the list); an entry is kept iff it owns its cell;
// kept entries are ranked in list order (stable: the first K + 2 points of the LIST build the locator, whatever the launch geometry) by
// a two-level count — 2048 entries per workgroup, ballots inside — and expanded to one (pos, src) pair per point.
constexpr uint32_t DED_THREADS = 256, DED_PER = 8, DED_CHUNK = DED_THREADS * DED_PER;

__global__ void dedup_claim_kernel(const uint32_t* __restrict__ idx, uint32_t n_cells, uint32_t domain_cells, uint32_t* __restrict__ owner,
                                   uint32_t* __restrict__ state) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_cells) return;
    const uint32_t ci = idx[r];
    if (ci >= domain_cells) {
        atomicOr(&state[1], 1u);  // an index outside the domain: reported by the host
        return;
    }
    atomicMin(&owner[ci], r);
}

__device__ __forceinline__ bool dedup_kept(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ owner, uint32_t r, uint32_t n_cells,
                                           uint32_t domain_cells) {
    if (r >= n_cells) return false;
    const uint32_t ci = idx[r];
    return ci < domain_cells && owner[ci] == r;
}

__global__ __launch_bounds__(DED_THREADS) void dedup_count_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ owner, uint32_t n_cells,
                                                                  uint32_t domain_cells, uint32_t* __restrict__ chunk_sum) {
    __shared__ uint32_t wave_sum[DED_THREADS / 64];
    const uint32_t base = blockIdx.x * DED_CHUNK;
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < DED_PER; j++) mine += dedup_kept(idx, owner, base + j * DED_THREADS + threadIdx.x, n_cells, domain_cells) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// exclusive prefix of the chunk sums (one workgroup, 1024 sums per round with a running carry); state[0] = the total
__global__ __launch_bounds__(1024) void dedup_offsets_kernel(const uint32_t* __restrict__ chunk_sum, uint32_t n_chunks, uint32_t* __restrict__ chunk_off,
                                                             uint32_t* __restrict__ state) {
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t at = 0; at < n_chunks; at += 1024) {
        const uint32_t i = at + threadIdx.x;
        const uint32_t v = i < n_chunks ? chunk_sum[i] : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
            const uint32_t add = threadIdx.x >= d ? buf[threadIdx.x - d] : 0u;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_chunks) chunk_off[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) state[0] = carry;
}

// kept entry of rank k: first_cell[k] = its cell, first_row[k] = its index in the caller's list
__global__ __launch_bounds__(DED_THREADS) void dedup_emit_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ owner, uint32_t n_cells,
                                                                 uint32_t domain_cells, const uint32_t* __restrict__ chunk_off,
                                                                 uint32_t* __restrict__ first_cell, uint32_t* __restrict__ first_row) {
    __shared__ uint32_t wave_cnt[DED_THREADS / 64];
    const uint32_t base = blockIdx.x * DED_CHUNK;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = chunk_off[blockIdx.x];
    for (uint32_t j = 0; j < DED_PER; j++) {
        const uint32_t r = base + j * DED_THREADS + threadIdx.x;
        const bool keep = dedup_kept(idx, owner, r, n_cells, domain_cells);
        const unsigned long long bal = __ballot(keep);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < DED_THREADS / 64; w++) {
            before += w < wave ? wave_cnt[w] : 0u;
            all += wave_cnt[w];
        }
        if (keep) {
            const uint32_t k = run + before + below;
            first_cell[k] = idx[r];
            first_row[k] = r;
        }
        run += all;
        __syncthreads();
    }
}

// point e of the kept cells: pos[e] = its position in the codeword, src[e] = its word offset in the caller's sample buffer
__global__ void dedup_expand_kernel(const uint32_t* __restrict__ first_cell, const uint32_t* __restrict__ first_row, const uint32_t* __restrict__ state,
                                    uint32_t ncols, uint32_t log_cell, uint32_t* __restrict__ pos, uint32_t* __restrict__ src) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ((size_t)state[0] << log_cell)) return;
    const uint32_t k = (uint32_t)(e >> log_cell), t = (uint32_t)e & ((1u << log_cell) - 1u);
    pos[e] = (first_cell[k] << log_cell) + t;
    src[e] = ((first_row[k] * ncols) << log_cell) + t;
}

__global__ void erasure_cell_firsts_kernel(const uint32_t* __restrict__ pos, uint32_t n_cells, uint32_t log_cell, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cells) out[i] = pos[(size_t)i << log_cell];
}

__global__ void erasure_points_kernel(ErasureDomain g, const uint32_t* __restrict__ pos, uint32_t count, uint32_t* __restrict__ px,
                                      uint32_t* __restrict__ py) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const CPoint p = domain_point(g, pos ? pos[t] : t);
    px[t] = p.x;
    py[t] = p.y;
}

// line a through the points at positions pos[2a] and pos[2a + 1]: A x + B y + C with (A, B, C) = (y1 - y2, x2 - x1, x1 y2 - x2 y1)
__global__ void erasure_lines_kernel(ErasureDomain g, const uint32_t* __restrict__ erased, uint32_t n_lines, uint32_t* __restrict__ la,
                                     uint32_t* __restrict__ lb, uint32_t* __restrict__ lc) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_lines) return;
    const CPoint p = domain_point(g, erased[2 * a]), q = domain_point(g, erased[2 * a + 1]);
    la[a] = m31_sub(p.y, q.y);
    lb[a] = m31_sub(q.x, p.x);
    lc[a] = m31_sub(m31_mul(p.x, q.y), m31_mul(q.x, p.y));
}

constexpr uint32_t Z_TILE = 512;  // lines staged through LDS per step

// zpart[chunk][t] = product over the lines of chunk `blockIdx.y` of line(P_t); a chunk = lines [chunk * per, (chunk + 1) * per).
// OWN: point t lies on line t >> 1 (the points are the pairs the lines were drawn through): that factor is the line's derivative
// along the circle's tangent at P_t, -y A + x B, instead of its value (zero).
template <bool OWN>
__global__ __launch_bounds__(256) void erasure_zeval_kernel(const uint32_t* __restrict__ px, const uint32_t* __restrict__ py, uint32_t count,
                                                            const uint32_t* __restrict__ la, const uint32_t* __restrict__ lb,
                                                            const uint32_t* __restrict__ lc, uint32_t n_lines, uint32_t per,
                                                            uint32_t* __restrict__ zpart) {
    __shared__ uint32_t sa[Z_TILE], sb[Z_TILE], sc[Z_TILE];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t x = t < count ? px[t] : 0u, y = t < count ? py[t] : 0u;
    const uint32_t ny = m31_neg(y);
    const uint32_t own = t >> 1;
    const uint32_t first = blockIdx.y * per;
    const uint32_t last = first + per < n_lines ? first + per : n_lines;
    uint32_t z = 1;
    for (uint32_t base = first; base < last; base += Z_TILE) {
        const uint32_t nt = last - base < Z_TILE ? last - base : Z_TILE;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) {
            sa[i] = la[base + i];
            sb[i] = lb[base + i];
            sc[i] = lc[base + i];
        }
        __syncthreads();
        for (uint32_t i = 0; i < nt; i++) {
            // A x + B y + C < 2 (P - 1)^2 + P < 2^63: one reduction
            uint64_t acc = (uint64_t)sa[i] * x + (uint64_t)sb[i] * y + sc[i];
            if (OWN && base + i == own) acc = (uint64_t)sa[i] * ny + (uint64_t)sb[i] * x;
            z = m31_mul(z, m31_reduce64(acc));
        }
    }
    if (t < count) zpart[(size_t)blockIdx.y * count + t] = z;
}

__global__ void erasure_zreduce_kernel(const uint32_t* __restrict__ zpart, uint32_t n_chunks, uint32_t count, uint32_t* __restrict__ z) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t v = 1;
    for (uint32_t c = 0; c < n_chunks; c++) v = m31_mul(v, zpart[(size_t)c * count + t]);
    z[t] = v;
}

// w[c][pos[t]] = z[t] * value of column c at known point t (value = cells[src[t] + c * 2^log_cell]); w was zeroed
__global__ void erasure_scatter_kernel(const uint32_t* __restrict__ cells, const uint32_t* __restrict__ src, const uint32_t* __restrict__ pos,
                                       const uint32_t* __restrict__ z, uint32_t count, uint32_t ncols, uint32_t log_cell, uint32_t* __restrict__ w,
                                       size_t w_stride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t zt = z[t];
    const size_t s0 = src[t];
    const size_t p = pos[t];
    for (uint32_t c = 0; c < ncols; c++) w[(size_t)c * w_stride + p] = m31_mul(cells[s0 + ((size_t)c << log_cell)], zt);
}

// weights of the known points: z[t] = V_D'(P_t) / Z_S'(P_t) with z[t] holding Z_S'(P_t) on entry; V_D' = -y prod_{j < n-1} 4 pi^j(x)
__global__ void erasure_known_weights_kernel(const uint32_t* __restrict__ px, const uint32_t* __restrict__ py, uint32_t count, uint32_t n,
                                             uint32_t* __restrict__ z) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t x = px[t], r = m31_neg(py[t]);
    for (uint32_t j = 0; j + 1 < n; j++) {
        const uint32_t x4 = m31_add(m31_add(x, x), m31_add(x, x));
        r = m31_mul(r, x4);
        x = double_x(x);
    }
    z[t] = m31_mul(r, m31_inv(z[t]));
}

// ---- samples that come as aligned cells of M = 2^m >= 2 entries: Z_S as a product over CELLS ----
// Cell c of the bit-reversed codeword (entries c M .. (c + 1) M) is a coset of the subgroup of order M / 2 together with its conjugates:
// m - 1 doublings take all of its points to ONE x-coordinate, so v_c(x) = pi^(m-1)(x) - k_c vanishes exactly on the cell (degree M / 2,
// M zeros).  With whole cells as S, Z_S = prod_c v_c costs one subtraction and one multiplication per (point, cell) instead of a line per
// (point, pair): K / M + 1 factors instead of K / 2 + 1.
__global__ void erasure_cellconst_kernel(ErasureDomain g, const uint32_t* __restrict__ cell_pos, uint32_t n_cells, uint32_t m, uint32_t* __restrict__ kc) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    uint32_t x = domain_point(g, cell_pos[c]).x;
    for (uint32_t j = 0; j + 1 < m; j++) x = double_x(x);
    kc[c] = x;
}

// zpart[chunk][t] = product over the cells of the chunk of (pi^(m-1)(x_t) - k_c); OWN: point t belongs to cell t >> m, whose factor
// (zero) is left out — erasure_known_weights_cells_kernel supplies its tangent derivative
template <bool OWN>
__global__ __launch_bounds__(256) void erasure_zeval_cells_kernel(const uint32_t* __restrict__ px, uint32_t count, uint32_t m,
                                                                  const uint32_t* __restrict__ kc, uint32_t n_cells, uint32_t per,
                                                                  uint32_t* __restrict__ zpart) {
    __shared__ uint32_t sk[3 * Z_TILE];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t u = t < count ? px[t] : 0u;
    for (uint32_t j = 0; j + 1 < m; j++) u = double_x(u);
    const uint32_t own = t >> m;
    const uint32_t first = blockIdx.y * per;
    const uint32_t last = first + per < n_cells ? first + per : n_cells;
    uint32_t z = 1;
    for (uint32_t base = first; base < last; base += 3 * Z_TILE) {
        const uint32_t nt = last - base < 3 * Z_TILE ? last - base : 3 * Z_TILE;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) sk[i] = kc[base + i];
        __syncthreads();
        for (uint32_t i = 0; i < nt; i++) {
            uint32_t f = m31_sub(u, sk[i]);
            if (OWN && base + i == own) f = 1u;
            z = m31_mul(z, f);
        }
    }
    if (t < count) zpart[(size_t)blockIdx.y * count + t] = z;
}

// z[t] = V_D'(P_t) / Z_S'(P_t) with z[t] holding the product over the OTHER cells on entry: the -y of the two tangent derivatives
// cancels and what is left of V_D' = -y prod_{j < n-1} 4 pi^j(x) over (pi^(m-1))' = -y prod_{j < m-1} 4 pi^j(x) is prod_{m-1 <= j < n-1} 4 pi^j(x)
__global__ void erasure_known_weights_cells_kernel(const uint32_t* __restrict__ px, uint32_t count, uint32_t n, uint32_t m, uint32_t* __restrict__ z) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t x = px[t], r = 1;
    for (uint32_t j = 0; j + 1 < n; j++) {
        if (j + 1 >= m) r = m31_mul(r, m31_add(m31_add(x, x), m31_add(x, x)));
        x = double_x(x);
    }
    z[t] = m31_mul(r, m31_inv(z[t]));
}

// block[c][t] = ev[c][t] * Z_S(P_t) / V_D(P_t) for the first `count` points P_t of D' (z[t] = Z_S(P_t), px = their x-coordinates)
__global__ void erasure_divide_kernel(const uint32_t* __restrict__ ev, size_t ev_stride, const uint32_t* __restrict__ z, const uint32_t* __restrict__ px,
                                      uint32_t count, uint32_t ncols, uint32_t n, uint32_t* __restrict__ block, size_t block_stride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t x = px[t];
    for (uint32_t j = 0; j + 1 < n; j++) x = double_x(x);  // V_D(P_t) = pi^(n-1)(x): non-zero off D
    const uint32_t f = m31_mul(z[t], m31_inv(x));
    for (uint32_t c = 0; c < ncols; c++) block[(size_t)c * block_stride + t] = m31_mul(ev[(size_t)c * ev_stride + t], f);
}

// ---- single points, large K: Z_S by a product tree instead of line by line ----
// Leaves of the tree: the product of 32 consecutive lines (degree 32) as its values on the canonic domain of 128 points (px, py):
// out[node][t], 128 threads per node, `nodes` leaves.  The few lines beyond 32 * nodes (Z_S has K / 2 + 1 lines, the tree K / 2) go into
// leaf 0: a node's domain has four points per line, so every node above leaf 0 has room for the extra degree.
__global__ __launch_bounds__(128) void erasure_lines32_kernel(const uint32_t* __restrict__ px, const uint32_t* __restrict__ py, const uint32_t* __restrict__ la,
                                                              const uint32_t* __restrict__ lb, const uint32_t* __restrict__ lc, uint32_t n_lines,
                                                              uint32_t nodes, uint32_t* __restrict__ out) {
    __shared__ uint32_t sa[32], sb[32], sc[32];
    const uint32_t node = blockIdx.x, t = threadIdx.x;
    const uint32_t first = 32u * node;
    if (t < 32) {
        const bool live = first + t < n_lines;
        sa[t] = live ? la[first + t] : 0u;
        sb[t] = live ? lb[first + t] : 0u;
        sc[t] = live ? lc[first + t] : 1u;
    }
    __syncthreads();
    const uint32_t x = px[t], y = py[t];
    uint32_t z = 1;
#pragma unroll 8
    for (int i = 0; i < 32; i++) z = m31_mul(z, m31_reduce64((uint64_t)sa[i] * x + (uint64_t)sb[i] * y + sc[i]));
    if (node == 0)
        for (uint32_t i = 32u * nodes; i < n_lines; i++) z = m31_mul(z, m31_reduce64((uint64_t)la[i] * x + (uint64_t)lb[i] * y + lc[i]));
    out[(size_t)node * 128 + t] = z;
}

// parent[i][t] = ext[2 i][t] * ext[2 i + 1][t] (a last child without a sibling is copied), t < size
__global__ void erasure_pairmul_kernel(const uint32_t* __restrict__ ext, uint32_t n_nodes, uint32_t size, uint32_t* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)((n_nodes + 1) / 2) * size;
    if (e >= total) return;
    const size_t i = e / size, t = e % size;
    const uint32_t a = ext[(2 * i) * size + t];
    out[e] = 2 * i + 1 < n_nodes ? m31_mul(a, ext[(2 * i + 1) * size + t]) : a;
}

// ze[t] = V_D(P_t) / zs[t] for the first `count` <= 2^n points P_t of the next canonic domain g1 (V_D = pi^(n-1)(x) of the domain of log size n)
// (a thread takes 8 points, lane-interleaved so that loads and stores stay coalesced, and inverts their product once: Montgomery's trick)
__global__ void erasure_ze_kernel(ErasureDomain g1, const uint32_t* __restrict__ zs, uint32_t count, uint32_t n, uint32_t* __restrict__ ze) {
    constexpr int PER = 8;
    const uint32_t t0 = blockIdx.x * blockDim.x * PER + threadIdx.x;
    uint32_t z[PER], pre[PER];
    uint32_t acc = 1;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t t = t0 + (uint32_t)j * blockDim.x;
        z[j] = t < count ? zs[t] : 1u;
        pre[j] = acc;  // product of z[0 .. j)
        acc = m31_mul(acc, z[j]);
    }
    uint32_t inv = m31_inv(acc);
    // V_D = pi^(n-1)(x) is constant on the first half of the next canonic domain: n - 1 doublings take its half coset (initial + j * step,
    // step of order 2^n) to initial' + j * (the point of order 2), i.e. to +-x' by the parity of j — and the first half of the
    // bit-reversed order is the even j
    uint32_t vd = domain_point(g1, 0).x;
    for (uint32_t k = 0; k + 1 < n; k++) vd = double_x(vd);
#pragma unroll
    for (int j = PER - 1; j >= 0; j--) {
        const uint32_t t = t0 + (uint32_t)j * blockDim.x;
        const uint32_t zi = m31_mul(inv, pre[j]);  // 1 / z[j]
        inv = m31_mul(inv, z[j]);
        if (t < count) ze[t] = m31_mul(vd, zi);
    }
}

__global__ void erasure_gather_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ pos, uint32_t count, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) out[t] = src[pos[t]];
}

// One step of "the first entries of a transform, without the transform": the first 2^out_log entries of the bit-reversed evaluation depend
// on the coefficient vector only through its fold along the layers above — at layer i (stride 2^i) an entry whose higher index bits are
// zero is v[j] + T_{i-1}[0] * v[j + 2^i] — so `cnt` <= 4 layers collapse 2^(out_log + cnt) coefficients per column into 2^out_log:
// out[c][j] = sum over m < 2^cnt of (product over the set bits b of m of T_{out_log + b - 1}[0]) * in[c][j + (m << out_log)].
__global__ void erasure_fold_prefix_kernel(const uint32_t* __restrict__ in, size_t in_stride, uint32_t out_log, uint32_t cnt, const uint32_t* __restrict__ tw,
                                           uint32_t dom_n, uint32_t* __restrict__ out, size_t out_stride) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ((size_t)1 << out_log)) return;
    const uint32_t* src = in + (size_t)blockIdx.y * in_stride + j;
    uint32_t t[4], val[16];
    for (uint32_t b = 0; b < cnt; b++) t[b] = tw[tw_level_offset_dev(dom_n, out_log + b - 1)];
    const uint32_t terms = 1u << cnt;
    for (uint32_t m = 0; m < terms; m++) val[m] = src[(size_t)m << out_log];
    for (uint32_t b = cnt; b-- > 0;)
        for (uint32_t m = 0; m < (1u << b); m++) val[m] = m31_add(val[m], m31_mul(t[b], val[m + (1u << b)]));
    out[(size_t)blockIdx.y * out_stride + j] = val[0];
}

// mismatch[0] += number of offered samples that differ from the re-encoded polynomial: ev[c][pos[t]] vs cells[src[t] + c * 2^log_cell]
__global__ void erasure_check_kernel(const uint32_t* __restrict__ cells, const uint32_t* __restrict__ src, const uint32_t* __restrict__ pos,
                                     uint32_t count, uint32_t ncols, uint32_t log_cell, const uint32_t* __restrict__ ev, size_t ev_stride,
                                     uint32_t* __restrict__ mismatch) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint32_t bad = 0;
    for (uint32_t c = 0; c < ncols; c++) bad += ev[(size_t)c * ev_stride + pos[t]] != cells[(size_t)src[t] + ((size_t)c << log_cell)];
    if (bad) atomicAdd(mismatch, bad);
}

}  // namespace

size_t erasure_zpart_chunks(uint32_t count, uint32_t n_lines) {
    if (n_lines == 0) return 1;
    // enough (point, chunk) pairs to fill the chip (~2^20 threads), at least one LDS tile of lines per chunk, at most 64 chunks
    size_t want = ((size_t)1 << 20) / (count ? count : 1);
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    const size_t max_by_lines = (n_lines + Z_TILE - 1) / Z_TILE;
    return want < max_by_lines ? want : max_by_lines;
}

size_t erasure_sample_lists_chunks(uint32_t n_cells) { return ((size_t)n_cells + DED_CHUNK - 1) / DED_CHUNK; }

hipError_t erasure_sample_lists(const Launch& L_, const uint32_t* d_idx, uint32_t n_cells, uint32_t domain_cells, uint32_t ncols, uint32_t log_cell,
                                uint32_t* d_owner, uint32_t* d_chunk_sum, uint32_t* d_chunk_off, uint32_t* d_first_cell, uint32_t* d_first_row,
                                uint32_t* d_state, uint32_t* d_pos, uint32_t* d_src) {
    if (!n_cells) return hipSuccess;
    Scope scope(L_, "erasure_sample_lists", 20.0 * n_cells + 8.0 * ((double)n_cells * (double)((size_t)1 << log_cell)));
    hipStream_t s = L_.stream;
    const unsigned chunks = (unsigned)erasure_sample_lists_chunks(n_cells);
    // (a failed memset would feed garbage into the claim and offset kernels: reported, not ignored)
    hipError_t e = hipMemsetAsync(d_owner, 0xFF, 4 * (size_t)domain_cells, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_state, 0, 8, s);
    if (e != hipSuccess) return e;
    dedup_claim_kernel<<<(n_cells + 255) / 256, 256, 0, s>>>(d_idx, n_cells, domain_cells, d_owner, d_state);
    dedup_count_kernel<<<chunks, DED_THREADS, 0, s>>>(d_idx, d_owner, n_cells, domain_cells, d_chunk_sum);
    dedup_offsets_kernel<<<1, 1024, 0, s>>>(d_chunk_sum, chunks, d_chunk_off, d_state);
    dedup_emit_kernel<<<chunks, DED_THREADS, 0, s>>>(d_idx, d_owner, n_cells, domain_cells, d_chunk_off, d_first_cell, d_first_row);
    const size_t pts_cap = (size_t)n_cells << log_cell;
    dedup_expand_kernel<<<(unsigned)((pts_cap + 255) / 256), 256, 0, s>>>(d_first_cell, d_first_row, d_state, ncols, log_cell, d_pos, d_src);
    return hipSuccess;
}

void erasure_cell_firsts(const Launch& L_, const uint32_t* d_pos, uint32_t n_cells, uint32_t log_cell, uint32_t* d_out) {
    if (!n_cells) return;
    erasure_cell_firsts_kernel<<<(n_cells + 255) / 256, 256, 0, L_.stream>>>(d_pos, n_cells, log_cell, d_out);
}

void erasure_points(const Launch& L_, const ErasureDomain& g, const uint32_t* d_pos, uint32_t count, uint32_t* d_px, uint32_t* d_py) {
    if (!count) return;
    Scope scope(L_, "erasure_points", 12.0 * count);
    erasure_points_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(g, d_pos, count, d_px, d_py);
}

void erasure_lines(const Launch& L_, const ErasureDomain& g, const uint32_t* d_erased, uint32_t n_lines, uint32_t* d_la, uint32_t* d_lb,
                   uint32_t* d_lc) {
    if (!n_lines) return;
    Scope scope(L_, "erasure_lines", 20.0 * n_lines);
    erasure_lines_kernel<<<(n_lines + 255) / 256, 256, 0, L_.stream>>>(g, d_erased, n_lines, d_la, d_lb, d_lc);
}

void erasure_zeval(const Launch& L_, const uint32_t* d_px, const uint32_t* d_py, uint32_t count, const uint32_t* d_la, const uint32_t* d_lb,
                   const uint32_t* d_lc, uint32_t n_lines, bool own, uint32_t* d_zpart, uint32_t* d_z) {
    if (!count) return;
    const uint32_t chunks = (uint32_t)erasure_zpart_chunks(count, n_lines);
    const uint32_t per = n_lines ? (n_lines + chunks - 1) / chunks : 0;
    {
        Scope scope(L_, "erasure_zeval", 8.0 * count + 12.0 * n_lines);
        const dim3 grid((count + 255) / 256, chunks);
        if (own)
            erasure_zeval_kernel<true><<<grid, 256, 0, L_.stream>>>(d_px, d_py, count, d_la, d_lb, d_lc, n_lines, per, d_zpart);
        else
            erasure_zeval_kernel<false><<<grid, 256, 0, L_.stream>>>(d_px, d_py, count, d_la, d_lb, d_lc, n_lines, per, d_zpart);
    }
    erasure_zreduce_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_zpart, chunks, count, d_z);
}

void erasure_cellconst(const Launch& L_, const ErasureDomain& g, const uint32_t* d_cell_pos, uint32_t n_cells, uint32_t m, uint32_t* d_kc) {
    if (!n_cells) return;
    erasure_cellconst_kernel<<<(n_cells + 255) / 256, 256, 0, L_.stream>>>(g, d_cell_pos, n_cells, m, d_kc);
}

void erasure_zeval_cells(const Launch& L_, const uint32_t* d_px, uint32_t count, uint32_t m, const uint32_t* d_kc, uint32_t n_cells, bool own,
                         uint32_t* d_zpart, uint32_t* d_z) {
    if (!count) return;
    const uint32_t chunks = (uint32_t)erasure_zpart_chunks(count, (n_cells + 2) / 3);
    const uint32_t per = n_cells ? (n_cells + chunks - 1) / chunks : 0;
    {
        Scope scope(L_, "erasure_zeval_cells", 8.0 * count + 4.0 * n_cells);
        const dim3 grid((count + 255) / 256, chunks);
        if (own)
            erasure_zeval_cells_kernel<true><<<grid, 256, 0, L_.stream>>>(d_px, count, m, d_kc, n_cells, per, d_zpart);
        else
            erasure_zeval_cells_kernel<false><<<grid, 256, 0, L_.stream>>>(d_px, count, m, d_kc, n_cells, per, d_zpart);
    }
    erasure_zreduce_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_zpart, chunks, count, d_z);
}

void erasure_known_weights_cells(const Launch& L_, const uint32_t* d_px, uint32_t count, uint32_t n, uint32_t m, uint32_t* d_z) {
    if (!count) return;
    erasure_known_weights_cells_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_px, count, n, m, d_z);
}

void erasure_lines32(const Launch& L_, const uint32_t* d_px128, const uint32_t* d_py128, const uint32_t* d_la, const uint32_t* d_lb, const uint32_t* d_lc,
                     uint32_t n_lines, uint32_t nodes, uint32_t* d_out) {
    if (!nodes) return;
    Scope scope(L_, "erasure_lines32", 12.0 * n_lines + 512.0 * nodes);
    erasure_lines32_kernel<<<nodes, 128, 0, L_.stream>>>(d_px128, d_py128, d_la, d_lb, d_lc, n_lines, nodes, d_out);
}

void erasure_pairmul(const Launch& L_, const uint32_t* d_ext, uint32_t n_nodes, uint32_t size, uint32_t* d_out) {
    const size_t total = (size_t)((n_nodes + 1) / 2) * size;
    if (!total) return;
    erasure_pairmul_kernel<<<(unsigned)((total + 255) / 256), 256, 0, L_.stream>>>(d_ext, n_nodes, size, d_out);
}

void erasure_ze(const Launch& L_, const ErasureDomain& g1, const uint32_t* d_zs, uint32_t count, uint32_t n, uint32_t* d_ze) {
    if (!count) return;
    Scope scope(L_, "erasure_ze", 8.0 * count);
    erasure_ze_kernel<<<(count + 2047) / 2048, 256, 0, L_.stream>>>(g1, d_zs, count, n, d_ze);
}

void erasure_fold_prefix(const Launch& L_, const uint32_t* d_in, size_t in_stride, uint32_t ncols, uint32_t out_log, uint32_t cnt, const uint32_t* d_tw,
                         uint32_t dom_n, uint32_t* d_out, size_t out_stride) {
    Scope scope(L_, "erasure_fold_prefix", 4.0 * ncols * (double)(((size_t)1 << (out_log + cnt)) + ((size_t)1 << out_log)));
    const size_t outs = (size_t)1 << out_log;
    dim3 grid((unsigned)((outs + 255) / 256), ncols);
    erasure_fold_prefix_kernel<<<grid, 256, 0, L_.stream>>>(d_in, in_stride, out_log, cnt, d_tw, dom_n, d_out, out_stride);
}

void erasure_gather(const Launch& L_, const uint32_t* d_src, const uint32_t* d_pos, uint32_t count, uint32_t* d_out) {
    if (!count) return;
    erasure_gather_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_src, d_pos, count, d_out);
}

void erasure_known_weights(const Launch& L_, const uint32_t* d_px, const uint32_t* d_py, uint32_t count, uint32_t n, uint32_t* d_z) {
    if (!count) return;
    erasure_known_weights_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_px, d_py, count, n, d_z);
}

void erasure_scatter(const Launch& L_, const uint32_t* d_cells, const uint32_t* d_src, const uint32_t* d_pos, const uint32_t* d_z, uint32_t count,
                     uint32_t ncols, uint32_t log_cell, uint32_t* d_w, size_t w_stride) {
    if (!count) return;
    Scope scope(L_, "erasure_scatter", (12.0 + 8.0 * ncols) * count);
    erasure_scatter_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_cells, d_src, d_pos, d_z, count, ncols, log_cell, d_w, w_stride);
}

void erasure_divide(const Launch& L_, const uint32_t* d_ev, size_t ev_stride, const uint32_t* d_z, const uint32_t* d_px, uint32_t count, uint32_t ncols,
                    uint32_t n, uint32_t* d_block, size_t block_stride) {
    if (!count) return;
    Scope scope(L_, "erasure_divide", (8.0 + 8.0 * ncols) * count);
    erasure_divide_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_ev, ev_stride, d_z, d_px, count, ncols, n, d_block, block_stride);
}

void erasure_check(const Launch& L_, const uint32_t* d_cells, const uint32_t* d_src, const uint32_t* d_pos, uint32_t count, uint32_t ncols,
                   uint32_t log_cell, const uint32_t* d_ev, size_t ev_stride, uint32_t* d_mismatch) {
    if (!count) return;
    Scope scope(L_, "erasure_check", (8.0 + 8.0 * ncols) * count);
    erasure_check_kernel<<<(count + 255) / 256, 256, 0, L_.stream>>>(d_cells, d_src, d_pos, count, ncols, log_cell, d_ev, ev_stride, d_mismatch);
}

}  // namespace k
}  // namespace frieda
