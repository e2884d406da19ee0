// clock_probe.hip — which clock does the chip hold under the integer load of the Merkle hash, and what does each opcode cost in
// CYCLES (not in wall time)?  DESIGN.md §5 needs both: a wall-time rate (lane-ops/s, compressions/s) mixes the issue cost with the
// clock the chip settles on under that load (MI355X_MICROARCH.md "DVFS give-back" (6): in-kernel clock = d(s_memtime) /
// d(s_memrealtime) x 100 MHz, after >= 2 s of back-to-back launches on non-trivial data).
//
// Every kernel: 2048 workgroups x 256 threads (8 waves per SIMD), every lane runs `iters` x UNROLL instructions of one opcode on 8
// independent register chains (or chained Blake2s compressions), wave 0 of a workgroup stamps both counters at its start and end
// into a buffer nothing else reads.  Launched back to back for ~2 s; the stamps of the last launch are reported:
//   clock     median over workgroups of d(memtime) / d(memrealtime) x 0.1 GHz
//   cyc/inst  SIMD cycles per wave-instruction = d(memtime) / (instructions per wave x 8 co-resident waves)   [median]
//   rate      wall-clock lane-ops/s (or compressions/s) of that last launch, HIP events
// Build: hipcc -O3 --offload-arch=gfx950 -Ifrieda_amd/csrc tools/clock_probe.hip -o tools/clock_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "blake2s.h"

using namespace frieda;

struct Stamp {
    unsigned long long c0, r0, c1, r1;
};

// Both counters in one volatile asm (s_memtime: shader clock; s_memrealtime: 100 MHz wall clock), waited for at once.  Volatile asm
// statements keep their order among themselves; the empty ones around them tie the stamp to the data flow of the loop it brackets
// (the loop's input depends on the first stamp, the second stamp follows an asm that consumes the loop's output), so the compiler
// can move neither across the loop.
__device__ __forceinline__ void stamp_pair(unsigned long long& c, unsigned long long& r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
}

#define STAMP_BEGIN(dep)                                  \
    unsigned long long c0_, r0_, c1_, r1_;                \
    stamp_pair(c0_, r0_);                                 \
    asm volatile("" : "+v"(dep) : "s"(c0_));
#define STAMP_END(st, dep)                                \
    asm volatile("" ::"v"(dep));                          \
    stamp_pair(c1_, r1_);                                 \
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0_, r0_, c1_, r1_};

constexpr int UNROLL = 64;  // instructions per loop iteration (8 chains x 8)

#define OP_KERNEL(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, Stamp* st, int iters) {           \
        uint32_t r[8], k = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;             \
        for (int i = 0; i < 8; i++) r[i] = k * (2 * i + 1) + 0x9E3779B9u * i;                    \
        uint32_t s1 = k ^ 0x5bd1e995u, s2 = (k >> 3) | 1u;                                       \
        STAMP_BEGIN(r[0])                                                                        \
        for (int it = 0; it < iters; it++) {                                                     \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; u++) {                              \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(r[i]) : "v"(s1), "v"(s2)); \
            }                                                                                    \
        }                                                                                        \
        STAMP_END(st, r[0])                                                                      \
        uint32_t s = 0;                                                                          \
        for (int i = 0; i < 8; i++) s ^= r[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                 \
    }

OP_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
OP_KERNEL(k_add, "v_add_u32 %0, %0, %1")
OP_KERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0")
OP_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %0, 7")
OP_KERNEL(k_alignbit16, "v_alignbit_b32 %0, %0, %0, 16")
OP_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
OP_KERNEL(k_perm, "v_perm_b32 %0, %0, %0, %2")
OP_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %2")
OP_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")

template <int LEAF>
__global__ __launch_bounds__(256) void k_blake(uint32_t* out, Stamp* st, int iters) {
    uint32_t m[16], h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    STAMP_BEGIN(m[0])
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
    STAMP_END(st, h[0])
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t*, Stamp*, int);

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

// inst_per_iter: wave-instructions of the opcode under test per loop iteration (Blake2s: 1 "instruction" = 1 compression)
static void run(const char* name, kern_t kfn, int iters, double inst_per_iter, int blocks_per_cu, double seconds, const char* unit) {
    const int blocks = 256 * blocks_per_cu;
    uint32_t* d_out;
    Stamp* d_st;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&d_st, sizeof(Stamp) * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; i++) kfn<<<blocks, 256>>>(d_out, d_st, iters);
        (void)hipDeviceSynchronize();
        launches += 8;
    }
    (void)hipEventRecord(e0);
    kfn<<<blocks, 256>>>(d_out, d_st, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(blocks);
    (void)hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk, cpi;
    for (const Stamp& s : st) {
        const double dc = (double)(s.c1 - s.c0), dr = (double)(s.r1 - s.r0);
        if (dr <= 0) continue;
        clk.push_back(dc / dr * 0.1);
        cpi.push_back(dc / ((double)iters * inst_per_iter * blocks_per_cu));  // waves per SIMD = blocks per CU (a block = 4 waves, one per SIMD)
    }
    const double total = (double)blocks * 256.0 * iters * inst_per_iter;
    printf("%-22s %d waves/SIMD  clock %5.3f GHz  %8.2f cycles per wave-%s per SIMD  rate %8.2f G lane-%s/s  (%.3f ms, after %d launches)\n", name,
           blocks_per_cu, median(clk), median(cpi), unit, total / (ms * 1e-3) / 1e9, unit, ms, launches);
    fflush(stdout);
    (void)hipFree(d_out);
    (void)hipFree(d_st);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    printf("# tools/clock_probe.hip: in-kernel clock (s_memtime / s_memrealtime) and cycles per instruction under sustained load, %g s per row\n", secs);
    run("v_xor_b32", k_xor, 4000, UNROLL, 8, secs, "inst");
    run("v_add_u32", k_add, 4000, UNROLL, 8, secs, "inst");
    run("v_lshrrev_b32", k_lshr, 4000, UNROLL, 8, secs, "inst");
    run("v_fma_f32", k_fma, 4000, UNROLL, 8, secs, "inst");
    run("v_alignbit_b32 (7)", k_alignbit, 4000, UNROLL, 8, secs, "inst");
    run("v_alignbit_b32 (16)", k_alignbit16, 4000, UNROLL, 8, secs, "inst");
    run("v_add3_u32", k_add3, 4000, UNROLL, 8, secs, "inst");
    run("v_perm_b32", k_perm, 4000, UNROLL, 8, secs, "inst");
    run("v_mul_lo_u32", k_mul_lo, 4000, UNROLL, 8, secs, "inst");
    for (int b : {1, 2, 4, 8}) run("blake2s node", k_blake<0>, 400, 1.0, b, secs, "compression");
    for (int b : {1, 4, 8}) run("blake2s leaf", k_blake<1>, 400, 1.0, b, secs, "compression");
    return 0;
}
