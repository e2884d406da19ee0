// codec.hip — byte stream -> M31 coefficient unpacker (gfx950).
//
// Replaces `bytes_to_felt_le` + the zero padding of `polynomial_from_bytes`
// (/root/reference/src/utils.rs:10-24): the input is an LSB-first bit stream cut into 30-bit chunks,
// the last chunk zero-extended, then zeros up to the padded power-of-two length.
//
// Layout: four felts are exactly 120 bits = 15 bytes, so thread t owns bytes [15t, 15t+15) and writes felts 4t..4t+3 as one
// 16-byte store.  A workgroup's 256 windows cover 3840 contiguous bytes = 960 aligned dwords: they are staged through LDS
// with coalesced 16-byte loads (dword loads at the ragged end of the blob), then every thread funnel-shifts its own 15-byte
// window out of five consecutive LDS words.
// HBM-bound: 3.75 B read + 4 B written per felt.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace frieda {
namespace k {

namespace {

// aligned dword `d` of the byte stream, zero beyond `len`
__device__ __forceinline__ uint32_t load_dword_guarded(const uint8_t* in, size_t len, size_t d) {
    size_t b = d * 4;
    if (b + 4 <= len) return *reinterpret_cast<const uint32_t*>(in + b);
    uint32_t v = 0;
    for (int i = 0; i < 4; i++)
        if (b + i < len) v |= (uint32_t)in[b + i] << (8 * i);
    return v;
}

template <int O>  // O = bit offset of the thread's first byte inside its first dword
__device__ __forceinline__ uint4 extract4(const uint32_t (&w)[5]) {
    uint32_t f[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pos = O + 30 * j, q = pos >> 5, r = pos & 31;
        uint64_t pair = ((uint64_t)w[q + 1] << 32) | w[q];  // q + 1 <= 4 for every (O, j)
        f[j] = (uint32_t)(pair >> r) & 0x3fffffffu;
    }
    return make_uint4(f[0], f[1], f[2], f[3]);
}

// T tiles of 256 quads per workgroup: T * 240 16-byte loads in flight per workgroup before anything waits (one tile per workgroup left
// the kernel at 0.9 TB/s: too few bytes in flight per CU for the memory latency)
template <int T>
__global__ __launch_bounds__(256) void unpack30_aligned_kernel(const uint8_t* __restrict__ in, size_t len,
                                                               uint32_t* __restrict__ out, size_t n_quads, size_t in_bstride,
                                                               size_t out_bstride) {
    in += blockIdx.y * in_bstride;  // blob of a batch
    out = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out) + blockIdx.y * out_bstride);
    // the workgroup's 256 T quads cover bytes [3840 T b, 3840 T (b + 1)) = 960 T aligned dwords: stage them through LDS with
    // coalesced loads, then every thread funnel-shifts its own 15-byte windows
    constexpr uint32_t DW = 960u * T;
    __shared__ __attribute__((aligned(16))) uint32_t stage[DW + 4];
    const size_t d_base = (size_t)blockIdx.x * DW;
    if ((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (d_base + DW + 4) * 4 <= len) {
        // interior workgroup: 240 T 16-byte loads (3840 = 16 * 240, so every workgroup's window starts 16-byte aligned) + 1 dword
        const uint4* src = reinterpret_cast<const uint4*>(in + 4 * d_base);
#pragma unroll
        for (uint32_t u = threadIdx.x; u < 240u * T; u += 256) reinterpret_cast<uint4*>(stage)[u] = src[u];
        if (threadIdx.x < 4) stage[DW + threadIdx.x] = reinterpret_cast<const uint32_t*>(in)[d_base + DW + threadIdx.x];
    } else {
        for (uint32_t i = threadIdx.x; i < DW + 4; i += 256) stage[i] = load_dword_guarded(in, len, d_base + i);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < T; i++) {
        const uint32_t local = (uint32_t)i * 256u + threadIdx.x;  // quad inside the workgroup's window
        const size_t t = (size_t)blockIdx.x * (256u * T) + local;
        if (t >= n_quads) return;
        const uint32_t byte0 = 15u * local;  // relative to the workgroup's first byte (a multiple of 4)
        const uint32_t d0 = byte0 >> 2;
        uint32_t w[5];
#pragma unroll
        for (int j = 0; j < 5; j++) w[j] = stage[d0 + j];
        uint4 v;
        switch (byte0 & 3) {
            case 0: v = extract4<0>(w); break;
            case 1: v = extract4<8>(w); break;
            case 2: v = extract4<16>(w); break;
            default: v = extract4<24>(w); break;
        }
        reinterpret_cast<uint4*>(out)[t] = v;
    }
}

// any alignment: one felt per thread from byte loads
__global__ __launch_bounds__(256) void unpack30_bytes_kernel(const uint8_t* __restrict__ in, size_t len,
                                                             uint32_t* __restrict__ out, size_t n_out, size_t in_bstride,
                                                             size_t out_bstride) {
    in += blockIdx.y * in_bstride;
    out = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out) + blockIdx.y * out_bstride);
    size_t kf = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (kf >= n_out) return;
    size_t bit0 = 30 * kf, byte0 = bit0 >> 3;
    uint32_t sh = (uint32_t)(bit0 & 7);
    uint64_t acc = 0;
    for (int i = 0; i < 5; i++)
        if (byte0 + i < len) acc |= (uint64_t)in[byte0 + i] << (8 * i);
    out[kf] = (uint32_t)(acc >> sh) & 0x3fffffffu;
}

}  // namespace

void unpack30(const Launch& L, const uint8_t* d_bytes, size_t len, uint32_t* d_out, size_t n_out, size_t src_bstride) {
    if (n_out == 0) return;
    hipStream_t s = L.stream;
    Scope scope(L, "unpack30", (double)len + 4.0 * (double)n_out);
    bool aligned = ((reinterpret_cast<uintptr_t>(d_bytes) & 3) == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0) &&
                   (n_out % 4 == 0) && (src_bstride % 4 == 0) && (L.bstride % 16 == 0);
    if (aligned) {
        size_t quads = n_out / 4;
        const int tiles = (int)L.tune->unpack_tiles;  // tuning knob: quads per workgroup / 256 (1, 2, 4 or 8)
        const int T = quads >= 8192 ? tiles : 1;  // small blobs: as many workgroups as there are
        dim3 grid((unsigned)((quads + 256 * (size_t)T - 1) / (256 * (size_t)T)), L.batch);
        if (T == 8)
            unpack30_aligned_kernel<8><<<grid, 256, 0, s>>>(d_bytes, len, d_out, quads, src_bstride, L.bstride);
        else if (T == 4)
            unpack30_aligned_kernel<4><<<grid, 256, 0, s>>>(d_bytes, len, d_out, quads, src_bstride, L.bstride);
        else if (T == 2)
            unpack30_aligned_kernel<2><<<grid, 256, 0, s>>>(d_bytes, len, d_out, quads, src_bstride, L.bstride);
        else
            unpack30_aligned_kernel<1><<<grid, 256, 0, s>>>(d_bytes, len, d_out, quads, src_bstride, L.bstride);
    } else {
        dim3 grid((unsigned)((n_out + 255) / 256), L.batch);
        unpack30_bytes_kernel<<<grid, 256, 0, s>>>(d_bytes, len, d_out, n_out, src_bstride, L.bstride);
    }
}

}  // namespace k
}  // namespace frieda
