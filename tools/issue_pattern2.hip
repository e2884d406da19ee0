// issue_pattern2.hip — round 5: one structured attack on the cycles per Blake2s compression (task 2).
// tools/issue_pattern.hip showed: fast class S (v_xor / v_add, VOP2) 2.33 SIMD cycles per wave-instruction, slow class C
// (v_alignbit / v_add3, VOP3 with three sources) 4.2, and ANY 50:50 mix 3.75 - 3.9 instead of the 3.26 a linear model gives.
// Questions here, every loop written as ONE asm block with explicit VGPR numbers so that the compiler can neither rewrite the
// stream nor move a register to another bank:
//   (a) rotr16(d ^ a) as two v_xor_b32_sdwa (word selects; UNUSED_PAD then UNUSED_PRESERVE) against v_xor + v_alignbit;
//   (b) v_perm_b32 for the 16- and 8-bit rotates;
//   (c) S : C mixes of 2:1, 3:1, 1:2, 1:3 — is the mixed cost a ratio effect or a constant;
//   (d) sources spread over the four VGPR banks (reg mod 4) against all sources in one bank;
//   (e) the encoding: v_xor_b32_e64 (VOP3 encoding of a 2-source op) against the 32-bit VOP2 form;
//   (f) phase-locked waves: 1024-thread workgroups (4 waves per SIMD) running S x16 then C x16 with a barrier per iteration,
//       so that the waves of a SIMD are in the same class at the same time.
// Reported: SIMD cycles per wave-instruction (wall rate x in-kernel clock), at 8 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_pattern2.hip -o tools/issue_pattern2.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Stamp {
    unsigned long long c0, r0, c1, r1;
};

__device__ __forceinline__ void stamp_pair(unsigned long long& c, unsigned long long& r) {
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
}

// Registers: chains v32..v47 (16 chains), sources v48..v63.  Bank of a VGPR = number mod 4.
#define CLOB                                                                                                                        \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
        "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s40", "s41", "scc"

#define INIT                                                                                                                      \
    "v_mov_b32 v32, %1\n v_add_u32 v33, 0x9E3779B9, v32\n v_add_u32 v34, 0x9E3779B9, v33\n v_add_u32 v35, 0x9E3779B9, v34\n"      \
    "v_add_u32 v36, 0x9E3779B9, v35\n v_add_u32 v37, 0x9E3779B9, v36\n v_add_u32 v38, 0x9E3779B9, v37\n v_add_u32 v39, 0x9E3779B9, v38\n" \
    "v_add_u32 v40, 0x9E3779B9, v39\n v_add_u32 v41, 0x9E3779B9, v40\n v_add_u32 v42, 0x9E3779B9, v41\n v_add_u32 v43, 0x9E3779B9, v42\n" \
    "v_add_u32 v44, 0x9E3779B9, v43\n v_add_u32 v45, 0x9E3779B9, v44\n v_add_u32 v46, 0x9E3779B9, v45\n v_add_u32 v47, 0x9E3779B9, v46\n" \
    "v_add_u32 v48, 0x85EBCA6B, v47\n v_add_u32 v49, 0x85EBCA6B, v48\n v_add_u32 v50, 0x85EBCA6B, v49\n v_add_u32 v51, 0x85EBCA6B, v50\n" \
    "v_add_u32 v52, 0x85EBCA6B, v51\n v_add_u32 v53, 0x85EBCA6B, v52\n v_add_u32 v54, 0x85EBCA6B, v53\n v_add_u32 v55, 0x85EBCA6B, v54\n" \
    "v_add_u32 v56, 0x85EBCA6B, v55\n v_add_u32 v57, 0x85EBCA6B, v56\n v_add_u32 v58, 0x85EBCA6B, v57\n v_add_u32 v59, 0x85EBCA6B, v58\n" \
    "v_add_u32 v60, 0x85EBCA6B, v59\n v_add_u32 v61, 0x85EBCA6B, v60\n v_add_u32 v62, 0x85EBCA6B, v61\n v_add_u32 v63, 0x85EBCA6B, v62\n" \
    "s_mov_b32 s40, 0x01000302\n s_mov_b32 s41, 0x00030201\n"

#define FOLD                                                                                                               \
    "v_xor_b32 v32, v32, v33\n v_xor_b32 v34, v34, v35\n v_xor_b32 v36, v36, v37\n v_xor_b32 v38, v38, v39\n"                \
    "v_xor_b32 v40, v40, v41\n v_xor_b32 v42, v42, v43\n v_xor_b32 v44, v44, v45\n v_xor_b32 v46, v46, v47\n"                \
    "v_xor_b32 v32, v32, v34\n v_xor_b32 v36, v36, v38\n v_xor_b32 v40, v40, v42\n v_xor_b32 v44, v44, v46\n"                \
    "v_xor_b32 v32, v32, v36\n v_xor_b32 v40, v40, v44\n v_xor_b32 %0, v32, v40\n"

// the loop runs inside the asm block (scalar counter): BODY is repeated 4 times per trip
#define AKERNEL(NAME, THREADS, BODY, NINST, BARRIER)                                                          \
    __global__ __launch_bounds__(THREADS) void NAME(uint32_t* out, Stamp* st, int iters) {                    \
        uint32_t k = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, res;                           \
        unsigned long long c0_, r0_, c1_, r1_;                                                                \
        stamp_pair(c0_, r0_);                                                                                 \
        asm volatile("" : "+v"(k) : "s"(c0_));                                                                \
        asm volatile(INIT "s_mov_b32 s42, %2\n"                                                               \
                          "L_loop_%=:\n" BODY BODY BODY BODY BARRIER                                          \
                          "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 L_loop_%=\n" FOLD     \
                     : "=v"(res)                                                                              \
                     : "v"(k), "s"(iters)                                                                     \
                     : CLOB, "s42");                                                                          \
        asm volatile("" ::"v"(res));                                                                          \
        stamp_pair(c1_, r1_);                                                                                 \
        if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0_, r0_, c1_, r1_};                                     \
        out[blockIdx.x * THREADS + threadIdx.x] = res;                                                        \
    }                                                                                                         \
    constexpr int NAME##_n = 4 * (NINST);                                                                     \
    constexpr int NAME##_t = THREADS;

// ---- building blocks (i = chain 0..15 -> v32+i; sources chosen per test) ----
// S: v_xor chain ^= src ; A: v_add ; C: alignbit chain, chain, chain, 7 ; C2: alignbit with two distinct regs ; D: add3
#define X(d, s) "v_xor_b32 v" #d ", v" #d ", v" #s "\n"
#define XE64(d, s) "v_xor_b32_e64 v" #d ", v" #d ", v" #s "\n"
#define AD(d, s) "v_add_u32 v" #d ", v" #d ", v" #s "\n"
#define RT(d, r) "v_alignbit_b32 v" #d ", v" #d ", v" #d ", " #r "\n"
#define RT2(d, s, r) "v_alignbit_b32 v" #d ", v" #d ", v" #s ", " #r "\n"
#define A3(d, s, t) "v_add3_u32 v" #d ", v" #d ", v" #s ", v" #t "\n"
#define PERM16(d) "v_perm_b32 v" #d ", v" #d ", v" #d ", s40\n"
#define PERM8(d) "v_perm_b32 v" #d ", v" #d ", v" #d ", s41\n"
// rotr16(d ^ a) into d: two SDWA xors through a temporary t:  t.hi = d.lo ^ a.lo (PAD zeroes t.lo) ; t.lo = d.hi ^ a.hi (PRESERVE keeps t.hi)
#define SDWA_ROT16(t, d, a)                                                                                         \
    "v_xor_b32_sdwa v" #t ", v" #d ", v" #a " dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n" \
    "v_xor_b32_sdwa v" #t ", v" #d ", v" #a " dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define SDWA_X(d, s) "v_xor_b32_sdwa v" #d ", v" #d ", v" #s " dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n"
#define SDWA_XP(d, s) "v_xor_b32_sdwa v" #d ", v" #d ", v" #s " dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"

// (ref) pure classes, 16 instructions per body
AKERNEL(k_s, 256, X(32, 49) X(33, 50) X(34, 51) X(35, 48) X(36, 53) X(37, 54) X(38, 55) X(39, 52) X(40, 57) X(41, 58) X(42, 59) X(43, 56) X(44, 61) X(45, 62) X(46, 63) X(47, 60), 16, "")
AKERNEL(k_c, 256, RT(32, 7) RT(33, 7) RT(34, 7) RT(35, 7) RT(36, 7) RT(37, 7) RT(38, 7) RT(39, 7) RT(40, 7) RT(41, 7) RT(42, 7) RT(43, 7) RT(44, 7) RT(45, 7) RT(46, 7) RT(47, 7), 16, "")
// (e) the VOP3 encoding of the same 2-source xor
AKERNEL(k_s_e64, 256, XE64(32, 49) XE64(33, 50) XE64(34, 51) XE64(35, 48) XE64(36, 53) XE64(37, 54) XE64(38, 55) XE64(39, 52) XE64(40, 57) XE64(41, 58) XE64(42, 59) XE64(43, 56) XE64(44, 61) XE64(45, 62) XE64(46, 63) XE64(47, 60), 16, "")
// S with both sources in the destination's bank / all in different banks (k_s above: dst bank b, src bank b+1)
AKERNEL(k_s_samebank, 256, X(32, 48) X(33, 49) X(34, 50) X(35, 51) X(36, 52) X(37, 53) X(38, 54) X(39, 55) X(40, 56) X(41, 57) X(42, 58) X(43, 59) X(44, 60) X(45, 61) X(46, 62) X(47, 63), 16, "")
// (d) add3 with three sources in ONE bank, and spread over three banks
AKERNEL(k_d_same, 256, A3(32, 48, 52) A3(33, 49, 53) A3(34, 50, 54) A3(35, 51, 55) A3(36, 52, 56) A3(37, 53, 57) A3(38, 54, 58) A3(39, 55, 59) A3(40, 56, 60) A3(41, 57, 61) A3(42, 58, 62) A3(43, 59, 63) A3(44, 60, 48) A3(45, 61, 49) A3(46, 62, 50) A3(47, 63, 51), 16, "")
AKERNEL(k_d_spread, 256, A3(32, 49, 54) A3(33, 50, 55) A3(34, 51, 52) A3(35, 48, 53) A3(36, 53, 58) A3(37, 54, 59) A3(38, 55, 56) A3(39, 52, 57) A3(40, 57, 62) A3(41, 58, 63) A3(42, 59, 60) A3(43, 56, 61) A3(44, 61, 50) A3(45, 62, 51) A3(46, 63, 48) A3(47, 60, 49), 16, "")
// alignbit with two DIFFERENT source registers (same bank / different banks)
AKERNEL(k_c2_same, 256, RT2(32, 48, 7) RT2(33, 49, 7) RT2(34, 50, 7) RT2(35, 51, 7) RT2(36, 52, 7) RT2(37, 53, 7) RT2(38, 54, 7) RT2(39, 55, 7) RT2(40, 56, 7) RT2(41, 57, 7) RT2(42, 58, 7) RT2(43, 59, 7) RT2(44, 60, 7) RT2(45, 61, 7) RT2(46, 62, 7) RT2(47, 63, 7), 16, "")
AKERNEL(k_c2_spread, 256, RT2(32, 49, 7) RT2(33, 50, 7) RT2(34, 51, 7) RT2(35, 48, 7) RT2(36, 53, 7) RT2(37, 54, 7) RT2(38, 55, 7) RT2(39, 52, 7) RT2(40, 57, 7) RT2(41, 58, 7) RT2(42, 59, 7) RT2(43, 56, 7) RT2(44, 61, 7) RT2(45, 62, 7) RT2(46, 63, 7) RT2(47, 60, 7), 16, "")
// (b) v_perm_b32 only (selector in an SGPR), both rotations
AKERNEL(k_perm, 256, PERM16(32) PERM8(33) PERM16(34) PERM8(35) PERM16(36) PERM8(37) PERM16(38) PERM8(39) PERM16(40) PERM8(41) PERM16(42) PERM8(43) PERM16(44) PERM8(45) PERM16(46) PERM8(47), 16, "")
// (a) SDWA xor only (PAD form, PRESERVE form)
AKERNEL(k_sdwa_pad, 256, SDWA_X(32, 49) SDWA_X(33, 50) SDWA_X(34, 51) SDWA_X(35, 48) SDWA_X(36, 53) SDWA_X(37, 54) SDWA_X(38, 55) SDWA_X(39, 52) SDWA_X(40, 57) SDWA_X(41, 58) SDWA_X(42, 59) SDWA_X(43, 56) SDWA_X(44, 61) SDWA_X(45, 62) SDWA_X(46, 63) SDWA_X(47, 60), 16, "")
AKERNEL(k_sdwa_pres, 256, SDWA_XP(32, 49) SDWA_XP(33, 50) SDWA_XP(34, 51) SDWA_XP(35, 48) SDWA_XP(36, 53) SDWA_XP(37, 54) SDWA_XP(38, 55) SDWA_XP(39, 52) SDWA_XP(40, 57) SDWA_XP(41, 58) SDWA_XP(42, 59) SDWA_XP(43, 56) SDWA_XP(44, 61) SDWA_XP(45, 62) SDWA_XP(46, 63) SDWA_XP(47, 60), 16, "")
// (a) "d = rotr16(d ^ a)" on 8 chains: xor + alignbit (S C)  vs  two SDWA xors into a temporary that becomes the new d
// (chains v32..v39 are d, v48..v55 are a; the temporaries v40..v47 swap roles with d every other use, as a renamer would)
AKERNEL(k_rot16_ref, 256, X(32, 49) RT(32, 16) X(33, 50) RT(33, 16) X(34, 51) RT(34, 16) X(35, 48) RT(35, 16) X(36, 53) RT(36, 16) X(37, 54) RT(37, 16) X(38, 55) RT(38, 16) X(39, 52) RT(39, 16), 16, "")
AKERNEL(k_rot16_sdwa, 256,
        SDWA_ROT16(40, 32, 49) SDWA_ROT16(41, 33, 50) SDWA_ROT16(42, 34, 51) SDWA_ROT16(43, 35, 48) SDWA_ROT16(44, 36, 53) SDWA_ROT16(45, 37, 54) SDWA_ROT16(46, 38, 55) SDWA_ROT16(47, 39, 52)
        , 16, "")
// rotr16 by v_perm after the xor (S + perm)
AKERNEL(k_rot16_perm, 256, X(32, 49) PERM16(32) X(33, 50) PERM16(33) X(34, 51) PERM16(34) X(35, 48) PERM16(35) X(36, 53) PERM16(36) X(37, 54) PERM16(37) X(38, 55) PERM16(38) X(39, 52) PERM16(39), 16, "")
// (c) ratio mixes on independent chains (12 instructions per body)
AKERNEL(k_s2c1, 256, X(32, 49) X(33, 50) RT(34, 7) X(35, 48) X(36, 53) RT(37, 7) X(38, 55) X(39, 52) RT(40, 7) X(41, 58) X(42, 59) RT(43, 7), 12, "")
AKERNEL(k_s3c1, 256, X(32, 49) X(33, 50) X(34, 51) RT(35, 7) X(36, 53) X(37, 54) X(38, 55) RT(39, 7) X(40, 57) X(41, 58) X(42, 59) RT(43, 7), 12, "")
AKERNEL(k_s1c1, 256, X(32, 49) RT(33, 7) X(34, 51) RT(35, 7) X(36, 53) RT(37, 7) X(38, 55) RT(39, 7) X(40, 57) RT(41, 7) X(42, 59) RT(43, 7), 12, "")
AKERNEL(k_s1c2, 256, X(32, 49) RT(33, 7) RT(34, 7) X(35, 48) RT(36, 7) RT(37, 7) X(38, 55) RT(39, 7) RT(40, 7) X(41, 58) RT(42, 7) RT(43, 7), 12, "")
AKERNEL(k_s1c3, 256, X(32, 49) RT(33, 7) RT(34, 7) RT(35, 7) X(36, 53) RT(37, 7) RT(38, 7) RT(39, 7) X(40, 57) RT(41, 7) RT(42, 7) RT(43, 7), 12, "")
AKERNEL(k_s5c1, 256, X(32, 49) X(33, 50) X(34, 51) X(35, 48) X(36, 53) RT(37, 7) X(38, 55) X(39, 52) X(40, 57) X(41, 58) X(42, 59) RT(43, 7), 12, "")
// the G mix with bank-spread sources for the add3s: D S C A S C D S C A S C (12)
AKERNEL(k_g_spread, 256, A3(32, 49, 54) X(33, 50) RT(34, 16) AD(35, 48) X(36, 53) RT(37, 12) A3(38, 55, 56) X(39, 52) RT(40, 8) AD(41, 58) X(42, 59) RT(43, 7), 12, "")
AKERNEL(k_g_same, 256, A3(32, 48, 52) X(33, 49) RT(34, 16) AD(35, 51) X(36, 52) RT(37, 12) A3(38, 54, 58) X(39, 55) RT(40, 8) AD(41, 57) X(42, 58) RT(43, 7), 12, "")
// the G mix with every add3 replaced by two adds (14: 8 S + ... ) and rot16 by SDWA: A A [sdwa sdwa] A S C A A S C A S C
AKERNEL(k_g_split, 256, AD(32, 49) AD(32, 54) X(33, 50) RT(34, 16) AD(35, 48) X(36, 53) RT(37, 12) AD(38, 55) AD(38, 56) X(39, 52) RT(40, 8) AD(41, 58) X(42, 59) RT(43, 7), 14, "")
// (f) phase-locked: S x16 then C x16 per body; 1024-thread workgroups with / without a barrier per trip (4 bodies = 128 instructions)
#define S16 X(32, 49) X(33, 50) X(34, 51) X(35, 48) X(36, 53) X(37, 54) X(38, 55) X(39, 52) X(40, 57) X(41, 58) X(42, 59) X(43, 56) X(44, 61) X(45, 62) X(46, 63) X(47, 60)
#define C16 RT(32, 7) RT(33, 7) RT(34, 7) RT(35, 7) RT(36, 7) RT(37, 7) RT(38, 7) RT(39, 7) RT(40, 7) RT(41, 7) RT(42, 7) RT(43, 7) RT(44, 7) RT(45, 7) RT(46, 7) RT(47, 7)
AKERNEL(k_run16_1024, 1024, S16 C16, 32, "")
AKERNEL(k_run16_1024_bar, 1024, S16 C16, 32, "s_barrier\n")
AKERNEL(k_run16_256, 256, S16 C16, 32, "")
// longer runs: S x64 C x64 per body
AKERNEL(k_run64_1024, 1024, S16 S16 S16 S16 C16 C16 C16 C16, 128, "")
AKERNEL(k_run64_1024_bar, 1024, S16 S16 S16 S16 C16 C16 C16 C16, 128, "s_barrier\n")
// fine interleave in a 1024-thread workgroup with a barrier (control: the barrier's own cost)
AKERNEL(k_scsc_1024_bar, 1024, X(32, 49) RT(33, 7) X(34, 51) RT(35, 7) X(36, 53) RT(37, 7) X(38, 55) RT(39, 7) X(40, 57) RT(41, 7) X(42, 59) RT(43, 7) X(44, 61) RT(45, 7) X(46, 63) RT(47, 7), 16, "s_barrier\n")
AKERNEL(k_s_1024_bar, 1024, S16, 16, "s_barrier\n")

typedef void (*kern_t)(uint32_t*, Stamp*, int);

static void run(const char* name, kern_t kfn, int n_inst, int threads, int waves_per_simd, double seconds) {
    const int waves_per_block = threads / 64;
    const int blocks_per_cu = waves_per_simd * 4 / waves_per_block;
    if (blocks_per_cu < 1) return;
    const int blocks = 256 * blocks_per_cu;
    const int iters = std::max(64, 256000 / n_inst);
    uint32_t* d_out;
    Stamp* d_st;
    (void)hipMalloc(&d_out, (size_t)blocks * threads * 4);
    (void)hipMalloc(&d_st, sizeof(Stamp) * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; i++) kfn<<<blocks, threads>>>(d_out, d_st, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(e0);
    kfn<<<blocks, threads>>>(d_out, d_st, iters);
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
    const double wave_inst = (double)blocks * waves_per_block * iters * n_inst;  // wave-instructions in the launch
    const double cyc = 1024.0 * clock * 1e9 * (ms * 1e-3) / wave_inst;
    printf("%-58s %d waves/SIMD  clock %5.3f GHz  %5.2f SIMD cycles per wave-instruction (wall)  %6.2f (in-kernel median wave)\n", name, waves_per_simd, clock,
           cyc, dur[dur.size() / 2] / ((double)iters * n_inst * waves_per_simd));
    fflush(stdout);
    (void)hipFree(d_out);
    (void)hipFree(d_st);
}

#define RUN(label, K) run(label, K, K##_n, K##_t, w, secs)

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.5;
    for (int w : {8, 4}) {
        RUN("S  v_xor_b32 (src bank = dst bank + 1)", k_s);
        RUN("S  v_xor_b32, source in the destination's bank", k_s_samebank);
        RUN("S  v_xor_b32_e64 (VOP3 encoding)", k_s_e64);
        RUN("C  v_alignbit_b32 v, v, v", k_c);
        RUN("C  v_alignbit_b32 two registers, one bank", k_c2_same);
        RUN("C  v_alignbit_b32 two registers, two banks", k_c2_spread);
        RUN("D  v_add3_u32, three sources in ONE bank", k_d_same);
        RUN("D  v_add3_u32, sources in three banks", k_d_spread);
        RUN("v_perm_b32 (selector in SGPR)", k_perm);
        RUN("v_xor_b32_sdwa WORD_1 <- WORD_0, UNUSED_PAD", k_sdwa_pad);
        RUN("v_xor_b32_sdwa WORD_0 <- WORD_1, UNUSED_PRESERVE", k_sdwa_pres);
        RUN("rotr16(d^a): v_xor + v_alignbit   (per instruction, 2 per rotate)", k_rot16_ref);
        RUN("rotr16(d^a): 2 x v_xor_b32_sdwa   (per instruction, 2 per rotate)", k_rot16_sdwa);
        RUN("rotr16(d^a): v_xor + v_perm_b32   (per instruction, 2 per rotate)", k_rot16_perm);
        RUN("mix S:C = 5:1", k_s5c1);
        RUN("mix S:C = 3:1", k_s3c1);
        RUN("mix S:C = 2:1", k_s2c1);
        RUN("mix S:C = 1:1", k_s1c1);
        RUN("mix S:C = 1:2", k_s1c2);
        RUN("mix S:C = 1:3", k_s1c3);
        RUN("G mix D S C A S C D S C A S C, add3 sources in 3 banks", k_g_spread);
        RUN("G mix, add3 sources in one bank", k_g_same);
        RUN("G mix, add3 split into 2 v_add (14 instructions)", k_g_split);
        RUN("S x16 C x16, 256-thread workgroups", k_run16_256);
        RUN("S x16 C x16, 1024-thread workgroups", k_run16_1024);
        RUN("S x16 C x16, 1024-thread workgroups, barrier / 128 instr", k_run16_1024_bar);
        RUN("S x64 C x64, 1024-thread workgroups", k_run64_1024);
        RUN("S x64 C x64, 1024-thread workgroups, barrier / 512 instr", k_run64_1024_bar);
        RUN("S C S C, 1024-thread workgroups, barrier / 64 instr", k_scsc_1024_bar);
        RUN("S only, 1024-thread workgroups, barrier / 64 instr", k_s_1024_bar);
    }
    return 0;
}
