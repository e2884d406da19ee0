// blake2s_rate.hip — pure-compute ceiling of the Blake2s compression on MI355X: every lane chains compressions on
// register-resident data (no memory traffic), at several occupancies.  Measurement aid; prints G compressions/s.
// Build: hipcc -O3 --offload-arch=gfx950 -Ifrieda_amd/csrc tools/blake2s_rate.hip -o tools/blake2s_rate.bin
#include <hip/hip_runtime.h>

#include <cstdio>

#include "blake2s.h"

using namespace frieda;

template <int LEAF>
__global__ __launch_bounds__(256) void chain_kernel(uint32_t* out, int iters) {
    uint32_t m[16], h[8];
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    if (LEAF)
        for (int i = 4; i < 16; i++) m[i] = 0;
    for (int it = 0; it < iters; it++) {
        if (LEAF) {
            const uint32_t mm[16] = {m[0], m[1], m[2], m[3], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            b2_merkle_block(mm, h);
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            b2_merkle_block(m, h);
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LEAF>
void run(int blocks_per_cu) {
    uint32_t* d;
    int blocks = 256 * blocks_per_cu;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    chain_kernel<LEAF><<<blocks, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    chain_kernel<LEAF><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * iters;
    printf("%s  %d blocks/CU (%d waves/SIMD): %7.2f G compressions/s  (%.3f ms)\n", LEAF ? "leaf" : "node", blocks_per_cu, blocks_per_cu, n / (ms * 1e-3) / 1e9, ms);
    (void)hipFree(d);
}

int main() {
    for (int b : {1, 2, 3, 4, 8}) run<0>(b);
    for (int b : {1, 2, 3, 4, 8}) run<1>(b);
    return 0;
}
