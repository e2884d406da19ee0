// column.hip — stwo `ColumnOps::bit_reverse_column` on device columns (gfx950).
//
// The reference builds its polynomials through `CirclePoly::<CpuBackend>::new` and `SecureCirclePoly` columns
// (/root/reference/src/utils.rs:21,28, src/proof.rs:47-52); a backend that replaces `CpuBackend` there must also provide the
// `ColumnOps<T>` trait: `bit_reverse_column(&mut Col<B, T>)` swaps v[i] and v[brev(i)] for every i < brev(i), in place
// (stwo core/utils.rs bit_reverse).  Columns are SoA: a SecureColumn is four M31 columns, so the secure variant is the same
// permutation applied to each of `ncols` columns `stride` words apart.
//
// Large columns (log_size >= 12) move 64 x 64 tiles: with the index split as (h : 6 | m : log-12 | l : 6) the element at
// (h, m, l) goes to (brev l, brev m, brev h), so the tile with middle bits m lands, transposed and with both tile coordinates
// bit-reversed, on the tile with middle bits brev m.  A workgroup owns the pair {m, brev m} (m <= brev m): both tiles are read
// with 256-byte rows, exchanged through LDS (row pitch 65 words: the transposed reads are conflict-free) and written back with
// 256-byte rows.  HBM-bound: 8 bytes per element, every access a full 256-byte row.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace frieda {
namespace k {

namespace {

constexpr uint32_t BR_T = 6;                 // tile edge = 64
constexpr uint32_t BR_PITCH = (1u << BR_T) + 1;

__global__ __launch_bounds__(256) void bit_reverse_tiles_kernel(uint32_t* __restrict__ cols, size_t stride, uint32_t log_size) {
    __shared__ uint32_t A[64 * BR_PITCH];
    __shared__ uint32_t B[64 * BR_PITCH];
    const uint32_t mid_bits = log_size - 2 * BR_T;
    const uint32_t m = blockIdx.x;
    const uint32_t mr = mid_bits ? bit_reverse(m, mid_bits) : 0u;
    if (mr < m) return;  // the pair is owned by the workgroup of the smaller index
    uint32_t* v = cols + (size_t)blockIdx.y * stride;
    const uint32_t t = threadIdx.x, c = t & 63, r0 = t >> 6;  // 4 rows per pass
    const uint32_t cr = bit_reverse(c, BR_T);
    auto at = [&](uint32_t h, uint32_t mm, uint32_t l) -> size_t { return ((size_t)h << (log_size - BR_T)) | ((size_t)mm << BR_T) | l; };
    for (uint32_t r = r0; r < 64; r += 4) {
        A[r * BR_PITCH + c] = v[at(r, m, c)];
        if (mr != m) B[r * BR_PITCH + c] = v[at(r, mr, c)];
    }
    __syncthreads();
    // destination tile (row r, column c) <- source tile (row brev c, column brev r)
    for (uint32_t r = r0; r < 64; r += 4) {
        const uint32_t rr = bit_reverse(r, BR_T);
        v[at(r, mr, c)] = A[cr * BR_PITCH + rr];
        if (mr != m) v[at(r, m, c)] = B[cr * BR_PITCH + rr];
    }
}

// small columns: one thread per index, swap when i < brev(i)
__global__ __launch_bounds__(256) void bit_reverse_small_kernel(uint32_t* __restrict__ cols, size_t stride, uint32_t log_size) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (1u << log_size)) return;
    uint32_t* v = cols + (size_t)blockIdx.y * stride;
    const uint32_t j = bit_reverse(i, log_size);
    if (i < j) {
        const uint32_t a = v[i], b = v[j];
        v[i] = b;
        v[j] = a;
    }
}

}  // namespace

void bit_reverse_columns(const Launch& L, uint32_t* d_cols, size_t stride, uint32_t ncols, uint32_t log_size) {
    if (log_size < 1 || ncols == 0) return;  // a column of one element is its own reversal
    Scope scope(L, "bit_reverse_column", 8.0 * (double)ncols * (double)((size_t)1 << log_size));
    if (log_size >= 2 * BR_T) {
        const dim3 grid(1u << (log_size - 2 * BR_T), ncols);
        bit_reverse_tiles_kernel<<<grid, 256, 0, L.stream>>>(d_cols, stride, log_size);
    } else {
        const dim3 grid(((1u << log_size) + 255) / 256, ncols);
        bit_reverse_small_kernel<<<grid, 256, 0, L.stream>>>(d_cols, stride, log_size);
    }
}

}  // namespace k
}  // namespace frieda
