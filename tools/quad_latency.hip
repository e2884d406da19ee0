// quad_latency.hip — what one dependent tree level costs in the latency-bound top of a Merkle tree on MI355X, by hashing
// scheme: one compression per lane, or one per quad of lanes (variants of the message fetch).  Also reports the shader clock
// the chip runs at during such a chain (s_memtime ticks per 100 MHz wall tick).  Measurement aid, not part of the product.
// Build: hipcc -O3 --offload-arch=gfx950 -Ifrieda_amd/csrc tools/quad_latency.hip -o tools/quad_latency.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "blake2s.h"

using namespace frieda;

constexpr uint32_t QS = 9;
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
constexpr int QROT1 = 0x39, QROT2 = 0x4E, QROT3 = 0x93;

struct QuadOffsets {
    uint32_t w[40];
};
// arithmetic selection (no table in memory): sigma index for (round, slot) of lane q
__device__ __forceinline__ void quad_offsets_init(QuadOffsets& o, uint32_t q) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int base = (k < 2 ? 0 : 8) + (k & 1);
            uint32_t idx = b2detail::SIGMA[r][base];
            idx = (q == 1) ? (uint32_t)b2detail::SIGMA[r][base + 2] : idx;
            idx = (q == 2) ? (uint32_t)b2detail::SIGMA[r][base + 4] : idx;
            idx = (q == 3) ? (uint32_t)b2detail::SIGMA[r][base + 6] : idx;
            o.w[4 * r + k] = 4u * (idx + (idx >> 3));
        }
    }
}

struct Quad2 {
    uint32_t lo, hi;
};

// V0: the product's scheme — fetch the 4 message words of a round at the top of the round
__device__ __forceinline__ Quad2 compress_v0(const uint32_t* msg, const QuadOffsets& o, uint32_t q) {
    uint32_t a = 0, b = 0, c = b2detail::IV[0], d = b2detail::IV[4];
    c = q == 1 ? b2detail::IV[1] : c, c = q == 2 ? b2detail::IV[2] : c, c = q == 3 ? b2detail::IV[3] : c;
    d = q == 1 ? b2detail::IV[5] : d, d = q == 2 ? b2detail::IV[6] : d, d = q == 3 ? b2detail::IV[7] : d;
    const char* mbase = reinterpret_cast<const char*>(msg);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t m0 = *reinterpret_cast<const uint32_t*>(mbase + o.w[4 * r]), m1 = *reinterpret_cast<const uint32_t*>(mbase + o.w[4 * r + 1]);
        const uint32_t m2 = *reinterpret_cast<const uint32_t*>(mbase + o.w[4 * r + 2]), m3 = *reinterpret_cast<const uint32_t*>(mbase + o.w[4 * r + 3]);
        FR_B2_G(a, b, c, d, m0, m1);
        b = quad_perm<QROT1>(b), c = quad_perm<QROT2>(c), d = quad_perm<QROT3>(d);
        FR_B2_G(a, b, c, d, m2, m3);
        b = quad_perm<QROT3>(b), c = quad_perm<QROT2>(c), d = quad_perm<QROT1>(d);
    }
    return {a ^ c, b ^ d};
}

// V1: all forty message words fetched up front (one wait), then pure ALU
__device__ __forceinline__ Quad2 compress_v1(const uint32_t* msg, const QuadOffsets& o, uint32_t q) {
    uint32_t a = 0, b = 0, c = b2detail::IV[0], d = b2detail::IV[4];
    c = q == 1 ? b2detail::IV[1] : c, c = q == 2 ? b2detail::IV[2] : c, c = q == 3 ? b2detail::IV[3] : c;
    d = q == 1 ? b2detail::IV[5] : d, d = q == 2 ? b2detail::IV[6] : d, d = q == 3 ? b2detail::IV[7] : d;
    const char* mbase = reinterpret_cast<const char*>(msg);
    uint32_t m[40];
#pragma unroll
    for (int i = 0; i < 40; i++) m[i] = *reinterpret_cast<const uint32_t*>(mbase + o.w[i]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 10; r++) {
        FR_B2_G(a, b, c, d, m[4 * r], m[4 * r + 1]);
        b = quad_perm<QROT1>(b), c = quad_perm<QROT2>(c), d = quad_perm<QROT3>(d);
        FR_B2_G(a, b, c, d, m[4 * r + 2], m[4 * r + 3]);
        b = quad_perm<QROT3>(b), c = quad_perm<QROT2>(c), d = quad_perm<QROT1>(d);
    }
    return {a ^ c, b ^ d};
}

// MODE 0: quad V0, 1: quad V1, 2: one compression per lane (SoA LDS)
template <int MODE>
__global__ __launch_bounds__(256) void chain_kernel(uint32_t* out, unsigned long long* clk, int levels, int active_quads) {
    __shared__ uint32_t Q[2][128 * QS + 16];
    __shared__ uint32_t S[2][8 * 132];
    const uint32_t t = threadIdx.x, q = t & 3;
    QuadOffsets qo;
    quad_offsets_init(qo, q);
    for (uint32_t i = t; i < 128 * QS + 16; i += blockDim.x) Q[0][i] = i * 2654435761u + blockIdx.x, Q[1][i] = 0;
    for (uint32_t i = t; i < 8 * 132; i += blockDim.x) S[0][i] = i * 2654435761u + blockIdx.x, S[1][i] = 0;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    int cur = 0;
    for (int l = 0; l < levels; l++) {
        if (MODE < 2) {
            const uint32_t j = t >> 2;
            if ((int)j < active_quads) {
                Quad2 v = MODE == 0 ? compress_v0(&Q[cur][2 * QS * j], qo, q) : compress_v1(&Q[cur][2 * QS * j], qo, q);
                Q[cur ^ 1][QS * j + q] = v.lo;
                Q[cur ^ 1][QS * j + 4 + q] = v.hi;
                Q[cur ^ 1][QS * (j + 64) + q] = v.hi;
                Q[cur ^ 1][QS * (j + 64) + 4 + q] = v.lo;
            }
        } else {
            if ((int)t < active_quads) {  // here: active lanes
                uint32_t m[16], h[8];
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    uint2 v = *reinterpret_cast<const uint2*>(&S[cur][w * 132 + 2 * (t & 63)]);
                    m[w] = v.x, m[8 + w] = v.y;
                }
                b2_merkle_block(m, h);
#pragma unroll
                for (int w = 0; w < 8; w++) S[cur ^ 1][w * 132 + (t & 127)] = h[w];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (t == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
    out[blockIdx.x * 256 + t] = Q[cur][t] + S[cur][t];
}

template <int MODE>
void run(const char* name, int blocks, int threads, int active) {
    uint32_t* d;
    unsigned long long* c;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&c, (size_t)blocks * 16);
    const int levels = 200;
    chain_kernel<MODE><<<blocks, threads>>>(d, c, 10, active);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    chain_kernel<MODE><<<blocks, threads>>>(d, c, levels, active);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), c, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / levels, ns = (double)h[1] * 10.0 / levels;
    printf("%-34s blocks %4d threads %3d active %3d: %7.0f ticks/level  %7.1f ns/level  (ticks/ns %.2f; event %.1f ns/level)\n", name, blocks, threads,
           active, cyc, ns, cyc / ns, ms * 1e6 / levels);
    (void)hipFree(d);
    (void)hipFree(c);
}

int main() {
    for (int blocks : {1, 256, 2048}) {
        run<0>("quad, per-round fetch (product)", blocks, 256, 64);
        run<1>("quad, all words fetched up front", blocks, 256, 64);
        run<0>("quad, per-round fetch, 1 wave", blocks, 64, 16);
        run<1>("quad, up front, 1 wave", blocks, 64, 16);
        run<0>("quad, per-round fetch, 1 quad", blocks, 64, 1);
        run<1>("quad, up front, 1 quad", blocks, 64, 1);
        run<2>("one compression per lane, 4 waves", blocks, 256, 256);
        run<2>("one compression per lane, 1 wave", blocks, 64, 64);
    }
    return 0;
}
