// issue_pattern.hip — does the ORDER of full-rate (S: v_xor/v_add) and half-rate (C: v_alignbit/v_add3) VALU instructions in a wave's
// stream change what they cost?  tools/clock_probe.hip shows S = 2.27 and C = 4.2 SIMD cycles per wave-instruction in isolation at a
// steady 2.39 GHz, yet one Blake2s compression (480 S + 480 C) costs 3950 cycles = 4.1 per instruction, not 3115.
// Patterns below repeat over 8 independent register chains; reported: SIMD cycles per wave-instruction (wall rate x in-kernel clock).
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_pattern.hip -o tools/issue_pattern.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

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

#define S_OP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s1))
#define A_OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s1))
#define C_OP(i) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(r[i]))
#define D_OP(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s1), "v"(s2))

#define KERNEL(NAME, BODY, NINST)                                                                 \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, Stamp* st, int iters) {            \
        uint32_t r[8], k = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;              \
        for (int i = 0; i < 8; i++) r[i] = k * (2 * i + 1) + 0x9E3779B9u * i;                     \
        uint32_t s1 = k ^ 0x5bd1e995u, s2 = (k >> 3) | 1u;                                        \
        unsigned long long c0_, r0_, c1_, r1_;                                                    \
        stamp_pair(c0_, r0_);                                                                     \
        asm volatile("" : "+v"(r[0]) : "s"(c0_));                                                 \
        for (int it = 0; it < iters; it++) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 4; u++) { BODY }                                 \
        }                                                                                         \
        asm volatile("" ::"v"(r[0]));                                                             \
        stamp_pair(c1_, r1_);                                                                     \
        if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0_, r0_, c1_, r1_};                         \
        uint32_t s = 0;                                                                           \
        for (int i = 0; i < 8; i++) s ^= r[i];                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
    }                                                                                             \
    constexpr int NAME##_n = 4 * (NINST);

// 16 instructions per body: 8 S + 8 C in different orders (independent chains 0..7)
KERNEL(k_scsc, S_OP(0); C_OP(1); S_OP(2); C_OP(3); S_OP(4); C_OP(5); S_OP(6); C_OP(7); S_OP(1); C_OP(0); S_OP(3); C_OP(2); S_OP(5); C_OP(4); S_OP(7); C_OP(6);, 16)
KERNEL(k_sscc, S_OP(0); S_OP(1); C_OP(2); C_OP(3); S_OP(4); S_OP(5); C_OP(6); C_OP(7); S_OP(2); S_OP(3); C_OP(0); C_OP(1); S_OP(6); S_OP(7); C_OP(4); C_OP(5);, 16)
KERNEL(k_s4c4, S_OP(0); S_OP(1); S_OP(2); S_OP(3); C_OP(4); C_OP(5); C_OP(6); C_OP(7); S_OP(4); S_OP(5); S_OP(6); S_OP(7); C_OP(0); C_OP(1); C_OP(2); C_OP(3);, 16)
KERNEL(k_s8c8, S_OP(0); S_OP(1); S_OP(2); S_OP(3); S_OP(4); S_OP(5); S_OP(6); S_OP(7); C_OP(0); C_OP(1); C_OP(2); C_OP(3); C_OP(4); C_OP(5); C_OP(6); C_OP(7);, 16)
KERNEL(k_ssss, S_OP(0); S_OP(1); S_OP(2); S_OP(3); S_OP(4); S_OP(5); S_OP(6); S_OP(7); S_OP(0); S_OP(1); S_OP(2); S_OP(3); S_OP(4); S_OP(5); S_OP(6); S_OP(7);, 16)
KERNEL(k_cccc, C_OP(0); C_OP(1); C_OP(2); C_OP(3); C_OP(4); C_OP(5); C_OP(6); C_OP(7); C_OP(0); C_OP(1); C_OP(2); C_OP(3); C_OP(4); C_OP(5); C_OP(6); C_OP(7);, 16)
// S A (two different full-rate opcodes alternating), and S D / A C mixes
KERNEL(k_sasa, S_OP(0); A_OP(1); S_OP(2); A_OP(3); S_OP(4); A_OP(5); S_OP(6); A_OP(7); S_OP(1); A_OP(0); S_OP(3); A_OP(2); S_OP(5); A_OP(4); S_OP(7); A_OP(6);, 16)
KERNEL(k_cdcd, C_OP(0); D_OP(1); C_OP(2); D_OP(3); C_OP(4); D_OP(5); C_OP(6); D_OP(7); C_OP(1); D_OP(0); C_OP(3); D_OP(2); C_OP(5); D_OP(4); C_OP(7); D_OP(6);, 16)
// the Blake2s G mix (D S C A S C D S C A S C) on independent chains, and the same 12 with the S/A paired (D C S A | S S C C ...)
KERNEL(k_gmix, D_OP(0); S_OP(1); C_OP(2); A_OP(3); S_OP(4); C_OP(5); D_OP(6); S_OP(7); C_OP(0); A_OP(1); S_OP(2); C_OP(3);, 12)
KERNEL(k_gpair, D_OP(0); C_OP(1); D_OP(2); C_OP(3); C_OP(4); C_OP(5); S_OP(6); A_OP(7); S_OP(0); S_OP(1); A_OP(2); S_OP(3);, 12)
// dependent chains: one chain of S C S C ... (latency-bound when alone; with 8 waves the SIMD interleaves waves)
KERNEL(k_dep_scsc, S_OP(0); C_OP(0); S_OP(0); C_OP(0); S_OP(0); C_OP(0); S_OP(0); C_OP(0); S_OP(0); C_OP(0); S_OP(0); C_OP(0); S_OP(0); C_OP(0); S_OP(0); C_OP(0);, 16)
KERNEL(k_dep_sscc, S_OP(0); S_OP(0); C_OP(0); C_OP(0); S_OP(0); S_OP(0); C_OP(0); C_OP(0); S_OP(0); S_OP(0); C_OP(0); C_OP(0); S_OP(0); S_OP(0); C_OP(0); C_OP(0);, 16)

// dependent chains of ONE opcode (what the latency of a lone quad-hash wave is made of)
KERNEL(k_dep_s, S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0); S_OP(0);, 16)
KERNEL(k_dep_a, A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0); A_OP(0);, 16)
KERNEL(k_dep_c, C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0); C_OP(0);, 16)
KERNEL(k_dep_d, D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0); D_OP(0);, 16)
// the G function as ONE dependent chain, and with each v_add3 split into two v_add
KERNEL(k_dep_g, D_OP(0); S_OP(0); C_OP(0); A_OP(0); S_OP(0); C_OP(0); D_OP(0); S_OP(0); C_OP(0); A_OP(0); S_OP(0); C_OP(0);, 12)
KERNEL(k_dep_g_split, A_OP(0); A_OP(0); S_OP(0); C_OP(0); A_OP(0); S_OP(0); C_OP(0); A_OP(0); A_OP(0); S_OP(0); C_OP(0); A_OP(0); S_OP(0); C_OP(0);, 14)

typedef void (*kern_t)(uint32_t*, Stamp*, int);

static void run(const char* name, kern_t kfn, int n_inst, int blocks_per_cu, double seconds) {
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    uint32_t* d_out;
    Stamp* d_st;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&d_st, sizeof(Stamp) * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; i++) kfn<<<blocks, 256>>>(d_out, d_st, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(e0);
    kfn<<<blocks, 256>>>(d_out, d_st, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st(blocks);
    (void)hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk, dur;
    for (const Stamp& s : st) {
        const double dc = (double)(s.c1 - s.c0), dr = (double)(s.r1 - s.r0);
        if (dr > 0) clk.push_back(dc / dr * 0.1), dur.push_back(dc);
    }
    std::sort(clk.begin(), clk.end());
    std::sort(dur.begin(), dur.end());
    const double clock = clk[clk.size() / 2];
    const double wave_inst = (double)blocks * 4.0 * iters * n_inst;  // wave-instructions in the launch
    const double cyc = 1024.0 * clock * 1e9 * (ms * 1e-3) / wave_inst;
    printf("%-44s %d waves/SIMD  clock %5.3f GHz  %5.2f SIMD cycles per wave-instruction (wall)  %6.2f (in-kernel, median wave: cycles / instructions / waves)\n", name,
           blocks_per_cu, clock, cyc, dur[dur.size() / 2] / ((double)iters * n_inst * blocks_per_cu));
    fflush(stdout);
    (void)hipFree(d_out);
    (void)hipFree(d_st);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    if (argc > 2) {  // latency mode: one wave per SIMD, dependent chains
        run("dependent v_xor_b32", k_dep_s, k_dep_s_n, 1, secs);
        run("dependent v_add_u32", k_dep_a, k_dep_a_n, 1, secs);
        run("dependent v_alignbit_b32", k_dep_c, k_dep_c_n, 1, secs);
        run("dependent v_add3_u32", k_dep_d, k_dep_d_n, 1, secs);
        run("dependent G chain (12 instructions)", k_dep_g, k_dep_g_n, 1, secs);
        run("dependent G chain, v_add3 as 2 x v_add (14)", k_dep_g_split, k_dep_g_split_n, 1, secs);
        return 0;
    }
    for (int w : {8, 4, 2, 1}) {
        run("S S S S  (xor only)", k_ssss, k_ssss_n, w, secs);
        run("C C C C  (alignbit only)", k_cccc, k_cccc_n, w, secs);
        run("S C S C  (alternating)", k_scsc, k_scsc_n, w, secs);
        run("S S C C", k_sscc, k_sscc_n, w, secs);
        run("S S S S C C C C", k_s4c4, k_s4c4_n, w, secs);
        run("S x8 C x8", k_s8c8, k_s8c8_n, w, secs);
        run("S A S A  (xor, add alternating)", k_sasa, k_sasa_n, w, secs);
        run("C D C D  (alignbit, add3 alternating)", k_cdcd, k_cdcd_n, w, secs);
        run("Blake2s G order  D S C A S C D S C A S C", k_gmix, k_gmix_n, w, secs);
        run("same 12, grouped  D C D C C C S A S S A S", k_gpair, k_gpair_n, w, secs);
        run("one dependent chain  S C S C", k_dep_scsc, k_dep_scsc_n, w, secs);
        run("one dependent chain  S S C C", k_dep_sscc, k_dep_sscc_n, w, secs);
    }
    return 0;
}
