// blake2s_asm.hip — round 6: the Merkle-shape compression as ONE hand-scheduled asm block (tools/gen_blake2s_asm.py) against the
// product's data-flow-pinned C++ form (blake2s.h b2_merkle_block<0xB000>), node and leaf shape, every lane chaining compressions on
// register-resident data.  Every variant is first checked against the plain form on the host.
// Build: python tools/gen_blake2s_asm.py --bench > tools/blake2s_asm_variants.h
//        hipcc -O3 --offload-arch=gfx950 -Ifrieda_amd/csrc tools/blake2s_asm.hip -o tools/blake2s_asm.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "blake2s.h"
#include "blake2s_asm_variants.h"

using namespace frieda;

struct Stamp {
    unsigned long long c0, r0, c1, r1;
};
__device__ __forceinline__ void stamp_pair(unsigned long long& c, unsigned long long& r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
}

__device__ __forceinline__ void prod_node(const uint32_t (&m)[16], uint32_t (&out)[8]) { b2_merkle_block<FRIEDA_B2_IDLE_NODE>(m, out); }
__device__ __forceinline__ void prod_leaf(const uint32_t (&m)[16], uint32_t (&out)[8]) { b2_merkle_block<FRIEDA_B2_IDLE_LEAF>(m, out); }

using fn_t = void (*)(const uint32_t (&)[16], uint32_t (&)[8]);

template <fn_t F, int LEAF>
__device__ __forceinline__ void chain(uint32_t (&m)[16], uint32_t (&h)[8], int iters) {
    for (int it = 0; it < iters; it++) {
        uint32_t mm[16];
        for (int i = 0; i < 16; i++) mm[i] = (LEAF && i >= 4) ? 0u : m[i];
        F(mm, h);
        if (LEAF) {
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
}

template <fn_t F, int LEAF, int WAVES>
__global__ __launch_bounds__(256, WAVES) void chain_kernel(uint32_t* out, int iters, Stamp* st) {
    uint32_t m[16], h[8];
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int i = 0; i < 8; i++) h[i] = 0;
    unsigned long long c0, r0, c1, r1;
    stamp_pair(c0, r0);
    asm volatile("" : "+v"(m[0]) : "s"(c0));
    chain<F, LEAF>(m, h, iters);
    asm volatile("" ::"v"(h[0]), "v"(h[7]));
    stamp_pair(c1, r1);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, r0, c1, r1};
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// correctness: the digest of thread 0 of block 0 after `iters` chained compressions, against the plain form on the host
static uint32_t host_chain(int leaf, int iters) {
    uint32_t m[16], h[8] = {0};
    for (int i = 0; i < 16; i++) m[i] = i * 40503u;
    for (int it = 0; it < iters; it++) {
        uint32_t mm[16];
        for (int i = 0; i < 16; i++) mm[i] = (leaf && i >= 4) ? 0u : m[i];
        b2_merkle_block_lat(mm, h);
        if (leaf) {
            for (int i = 0; i < 4; i++) m[i] = h[i] ^ h[4 + i];
        } else {
            for (int i = 0; i < 8; i++) {
                m[i] ^= h[i];
                m[8 + i] += h[i];
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    return s;
}

typedef void (*kern_t)(uint32_t*, int, Stamp*);
static double run(const char* name, kern_t kfn, int leaf, int waves, double seconds) {
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 256, 0);
    const int per_cu = std::min(occ, waves);
    const int blocks = 256 * per_cu, iters = 96;
    uint32_t* d_out;
    Stamp* d_st;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&d_st, sizeof(Stamp) * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; i++) kfn<<<blocks, 256>>>(d_out, iters, d_st);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(e0);
    kfn<<<blocks, 256>>>(d_out, iters, d_st);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(blocks);
    (void)hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (const Stamp& s : st) {
        const double dc = (double)(s.c1 - s.c0), dr = (double)(s.r1 - s.r0);
        if (dr > 0) clk.push_back(dc / dr * 0.1);
    }
    std::sort(clk.begin(), clk.end());
    const double clock = clk[clk.size() / 2];
    const double rate = (double)blocks * 256.0 * iters / (ms * 1e-3);
    uint32_t first = 0;
    (void)hipMemcpy(&first, d_out, 4, hipMemcpyDeviceToHost);
    const bool ok = first == host_chain(leaf, iters);
    printf("%-44s %d waves/SIMD  clock %5.3f GHz  %6.2f G compressions/s  %7.1f SIMD cycles per wave-compression  %s\n", name, per_cu, clock, rate * 1e-9,
           1024.0 * 64.0 * clock * 1e9 / rate, ok ? "digest ok" : "DIGEST MISMATCH");
    fflush(stdout);
    (void)hipFree(d_out);
    (void)hipFree(d_st);
    return rate;
}

#define RUN(label, F, LEAF, W) run(label, chain_kernel<F, LEAF, W>, LEAF, W, secs)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.5;
    const int reps = argc > 2 ? atoi(argv[2]) : 2;
    for (int rep = 0; rep < reps; rep++) {
        RUN("node product (pinned C++, 0xB000)", prod_node, 0, 8);
        RUN("node asm block, rot 3 / add3 2 / fast 0", b2_asm_node, 0, 8);
        RUN("node asm block, slow 2 / fast 0", b2_asm_node_p2, 0, 8);
        RUN("node asm block, rot 3 / add3 1 / fast 0", b2_asm_node_p31, 0, 8);
        RUN("node asm block, prio per instruction class", b2_asm_node_cls, 0, 8);
        RUN("node asm block, rot16 as v_pk_add_u16", b2_asm_node_pk, 0, 8);
        RUN("node asm block, no priorities", b2_asm_node_nop, 0, 8);
        RUN("leaf product (pinned C++, 0xB000)", prod_leaf, 1, 8);
        RUN("leaf asm block, rot 3 / add3 2 / fast 0", b2_asm_leaf, 1, 8);
        RUN("leaf asm block, slow 2 / fast 0", b2_asm_leaf_p2, 1, 8);
        RUN("leaf asm block, rot 3 / add3 1 / fast 0", b2_asm_leaf_p31, 1, 8);
        RUN("leaf asm block, prio per instruction class", b2_asm_leaf_cls, 1, 8);
        RUN("leaf asm block, add3 before add in a run", b2_asm_leaf_split, 1, 8);
        RUN("leaf asm block, rot16 as v_pk_add_u16", b2_asm_leaf_pk, 1, 8);
        RUN("leaf asm block, no priorities", b2_asm_leaf_nop, 1, 8);
    }
    RUN("node product, 4 waves", prod_node, 0, 4);
    RUN("node asm block, 4 waves", b2_asm_node, 0, 4);
    RUN("leaf product, 4 waves", prod_leaf, 1, 4);
    RUN("leaf asm block, 4 waves", b2_asm_leaf, 1, 4);
    return 0;
}
