// issue_overlap.hip — round 5: do two VALU rate classes OVERLAP when they come from different waves of a SIMD?
// 512-thread workgroups (8 waves: wave w on SIMD w mod 4): waves 0 - 3 run stream X, waves 4 - 7 stream Y, 4 workgroups per CU, so
// every SIMD holds four X waves and four Y waves.  Each wave stamps its own start and end; reported per class: SIMD cycles per
// wave-instruction of that class (its waves' median duration / instructions) and what the two classes cost alone at 4 waves per SIMD.
// Streams: S v_xor, A v_add, C v_alignbit, D v_add3, M v_mad_u64_u32, N v_min_u32, L v_mul_lo_u32.  Optional: class X at s_setprio P.
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_overlap.hip -o tools/issue_overlap.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Stamp { unsigned long long c0, c1; };
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","s42","s44","s45","scc","vcc"
#define INIT_ \
 "v_mov_b32 v32, %1\n v_add_u32 v33, 0x9E3779B9, v32\n v_add_u32 v34, 0x9E3779B9, v33\n v_add_u32 v35, 0x9E3779B9, v34\n" \
 "v_add_u32 v36, 0x9E3779B9, v35\n v_add_u32 v37, 0x9E3779B9, v36\n v_add_u32 v38, 0x9E3779B9, v37\n v_add_u32 v39, 0x9E3779B9, v38\n" \
 "v_add_u32 v40, 0x9E3779B9, v39\n v_add_u32 v41, 0x9E3779B9, v40\n v_add_u32 v42, 0x9E3779B9, v41\n v_add_u32 v43, 0x9E3779B9, v42\n" \
 "v_add_u32 v44, 0x9E3779B9, v43\n v_add_u32 v45, 0x9E3779B9, v44\n v_add_u32 v46, 0x9E3779B9, v45\n v_add_u32 v47, 0x9E3779B9, v46\n" \
 "v_add_u32 v48, 0x85EBCA6B, v47\n v_add_u32 v49, 0x85EBCA6B, v48\n v_add_u32 v50, 0x85EBCA6B, v49\n v_add_u32 v51, 0x85EBCA6B, v50\n" \
 "v_add_u32 v52, 0x85EBCA6B, v51\n v_add_u32 v53, 0x85EBCA6B, v52\n v_add_u32 v54, 0x85EBCA6B, v53\n v_add_u32 v55, 0x85EBCA6B, v54\n" \
 "v_add_u32 v56, 0x85EBCA6B, v55\n v_add_u32 v57, 0x85EBCA6B, v56\n v_add_u32 v58, 0x85EBCA6B, v57\n v_add_u32 v59, 0x85EBCA6B, v58\n" \
 "v_add_u32 v60, 0x85EBCA6B, v59\n v_add_u32 v61, 0x85EBCA6B, v60\n v_add_u32 v62, 0x85EBCA6B, v61\n v_add_u32 v63, 0x85EBCA6B, v62\n"
#define FOLD_ \
 "v_xor_b32 v32, v32, v33\n v_xor_b32 v34, v34, v35\n v_xor_b32 v36, v36, v37\n v_xor_b32 v38, v38, v39\n" \
 "v_xor_b32 v40, v40, v41\n v_xor_b32 v42, v42, v43\n v_xor_b32 v44, v44, v45\n v_xor_b32 v46, v46, v47\n" \
 "v_xor_b32 v32, v32, v34\n v_xor_b32 v36, v36, v38\n v_xor_b32 v40, v40, v42\n v_xor_b32 v44, v44, v46\n" \
 "v_xor_b32 v32, v32, v36\n v_xor_b32 v40, v40, v44\n v_xor_b32 %0, v32, v40\n"
#define BODY_S \
 "v_xor_b32 v32, v32, v48\n" \
 "v_xor_b32 v33, v33, v49\n" \
 "v_xor_b32 v34, v34, v50\n" \
 "v_xor_b32 v35, v35, v51\n" \
 "v_xor_b32 v36, v36, v52\n" \
 "v_xor_b32 v37, v37, v53\n" \
 "v_xor_b32 v38, v38, v54\n" \
 "v_xor_b32 v39, v39, v55\n" \
 "v_xor_b32 v40, v40, v56\n" \
 "v_xor_b32 v41, v41, v57\n" \
 "v_xor_b32 v42, v42, v58\n" \
 "v_xor_b32 v43, v43, v59\n" \
 "v_xor_b32 v44, v44, v60\n" \
 "v_xor_b32 v45, v45, v61\n" \
 "v_xor_b32 v46, v46, v62\n" \
 "v_xor_b32 v47, v47, v63\n" \

#define BODY_A \
 "v_add_u32 v32, v32, v48\n" \
 "v_add_u32 v33, v33, v49\n" \
 "v_add_u32 v34, v34, v50\n" \
 "v_add_u32 v35, v35, v51\n" \
 "v_add_u32 v36, v36, v52\n" \
 "v_add_u32 v37, v37, v53\n" \
 "v_add_u32 v38, v38, v54\n" \
 "v_add_u32 v39, v39, v55\n" \
 "v_add_u32 v40, v40, v56\n" \
 "v_add_u32 v41, v41, v57\n" \
 "v_add_u32 v42, v42, v58\n" \
 "v_add_u32 v43, v43, v59\n" \
 "v_add_u32 v44, v44, v60\n" \
 "v_add_u32 v45, v45, v61\n" \
 "v_add_u32 v46, v46, v62\n" \
 "v_add_u32 v47, v47, v63\n" \

#define BODY_C \
 "v_alignbit_b32 v32, v32, v32, 7\n" \
 "v_alignbit_b32 v33, v33, v33, 7\n" \
 "v_alignbit_b32 v34, v34, v34, 7\n" \
 "v_alignbit_b32 v35, v35, v35, 7\n" \
 "v_alignbit_b32 v36, v36, v36, 7\n" \
 "v_alignbit_b32 v37, v37, v37, 7\n" \
 "v_alignbit_b32 v38, v38, v38, 7\n" \
 "v_alignbit_b32 v39, v39, v39, 7\n" \
 "v_alignbit_b32 v40, v40, v40, 7\n" \
 "v_alignbit_b32 v41, v41, v41, 7\n" \
 "v_alignbit_b32 v42, v42, v42, 7\n" \
 "v_alignbit_b32 v43, v43, v43, 7\n" \
 "v_alignbit_b32 v44, v44, v44, 7\n" \
 "v_alignbit_b32 v45, v45, v45, 7\n" \
 "v_alignbit_b32 v46, v46, v46, 7\n" \
 "v_alignbit_b32 v47, v47, v47, 7\n" \

#define BODY_D \
 "v_add3_u32 v32, v32, v48, v49\n" \
 "v_add3_u32 v33, v33, v49, v50\n" \
 "v_add3_u32 v34, v34, v50, v51\n" \
 "v_add3_u32 v35, v35, v51, v52\n" \
 "v_add3_u32 v36, v36, v52, v53\n" \
 "v_add3_u32 v37, v37, v53, v54\n" \
 "v_add3_u32 v38, v38, v54, v55\n" \
 "v_add3_u32 v39, v39, v55, v56\n" \
 "v_add3_u32 v40, v40, v56, v57\n" \
 "v_add3_u32 v41, v41, v57, v58\n" \
 "v_add3_u32 v42, v42, v58, v59\n" \
 "v_add3_u32 v43, v43, v59, v60\n" \
 "v_add3_u32 v44, v44, v60, v61\n" \
 "v_add3_u32 v45, v45, v61, v62\n" \
 "v_add3_u32 v46, v46, v62, v63\n" \
 "v_add3_u32 v47, v47, v63, v48\n" \

#define BODY_M \
 "v_mad_u64_u32 v[32:33], s[44:45], v48, v32, v[32:33]\n" \
 "v_mad_u64_u32 v[34:35], s[44:45], v49, v34, v[34:35]\n" \
 "v_mad_u64_u32 v[36:37], s[44:45], v50, v36, v[36:37]\n" \
 "v_mad_u64_u32 v[38:39], s[44:45], v51, v38, v[38:39]\n" \
 "v_mad_u64_u32 v[40:41], s[44:45], v52, v40, v[40:41]\n" \
 "v_mad_u64_u32 v[42:43], s[44:45], v53, v42, v[42:43]\n" \
 "v_mad_u64_u32 v[44:45], s[44:45], v54, v44, v[44:45]\n" \
 "v_mad_u64_u32 v[46:47], s[44:45], v55, v46, v[46:47]\n" \
 "v_mad_u64_u32 v[32:33], s[44:45], v56, v32, v[32:33]\n" \
 "v_mad_u64_u32 v[34:35], s[44:45], v57, v34, v[34:35]\n" \
 "v_mad_u64_u32 v[36:37], s[44:45], v58, v36, v[36:37]\n" \
 "v_mad_u64_u32 v[38:39], s[44:45], v59, v38, v[38:39]\n" \
 "v_mad_u64_u32 v[40:41], s[44:45], v60, v40, v[40:41]\n" \
 "v_mad_u64_u32 v[42:43], s[44:45], v61, v42, v[42:43]\n" \
 "v_mad_u64_u32 v[44:45], s[44:45], v62, v44, v[44:45]\n" \
 "v_mad_u64_u32 v[46:47], s[44:45], v63, v46, v[46:47]\n" \

#define BODY_N \
 "v_min_u32 v32, v32, v48\n" \
 "v_min_u32 v33, v33, v49\n" \
 "v_min_u32 v34, v34, v50\n" \
 "v_min_u32 v35, v35, v51\n" \
 "v_min_u32 v36, v36, v52\n" \
 "v_min_u32 v37, v37, v53\n" \
 "v_min_u32 v38, v38, v54\n" \
 "v_min_u32 v39, v39, v55\n" \
 "v_min_u32 v40, v40, v56\n" \
 "v_min_u32 v41, v41, v57\n" \
 "v_min_u32 v42, v42, v58\n" \
 "v_min_u32 v43, v43, v59\n" \
 "v_min_u32 v44, v44, v60\n" \
 "v_min_u32 v45, v45, v61\n" \
 "v_min_u32 v46, v46, v62\n" \
 "v_min_u32 v47, v47, v63\n" \

#define BODY_L \
 "v_mul_lo_u32 v32, v32, v48\n" \
 "v_mul_lo_u32 v33, v33, v49\n" \
 "v_mul_lo_u32 v34, v34, v50\n" \
 "v_mul_lo_u32 v35, v35, v51\n" \
 "v_mul_lo_u32 v36, v36, v52\n" \
 "v_mul_lo_u32 v37, v37, v53\n" \
 "v_mul_lo_u32 v38, v38, v54\n" \
 "v_mul_lo_u32 v39, v39, v55\n" \
 "v_mul_lo_u32 v40, v40, v56\n" \
 "v_mul_lo_u32 v41, v41, v57\n" \
 "v_mul_lo_u32 v42, v42, v58\n" \
 "v_mul_lo_u32 v43, v43, v59\n" \
 "v_mul_lo_u32 v44, v44, v60\n" \
 "v_mul_lo_u32 v45, v45, v61\n" \
 "v_mul_lo_u32 v46, v46, v62\n" \
 "v_mul_lo_u32 v47, v47, v63\n" \


// one stream: the loop runs inside the asm block, BODY 4 times per trip (64 instructions)
#define STREAM(BODY, PRIO) \
    asm volatile("s_setprio " #PRIO "\n" INIT_ "s_mov_b32 s42, %2\n s_mov_b32 s44, 0x12345\n s_mov_b32 s45, 0\n" \
                 "1:\n" BODY BODY BODY BODY "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 1b\n" FOLD_ "s_setprio 0\n" \
                 : "=v"(res) : "v"(k), "s"(iters) : CLOB)
#define PAIR(NAME, BX, BY, PX)                                                                          \
    __global__ __launch_bounds__(512) void NAME(uint32_t* out, Stamp* st, int iters_x, int iters_y) {   \
        uint32_t k = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, res = 0;                 \
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                              \
        unsigned long long c0 = __builtin_readcyclecounter();                                           \
        asm volatile("" : "+v"(k) : "s"(c0));                                                           \
        if (wave < 4) {                                                                                 \
            const int iters = iters_x;                                                                  \
            if (iters > 0) STREAM(BX, PX);                                                              \
        } else {                                                                                        \
            const int iters = iters_y;                                                                  \
            if (iters > 0) STREAM(BY, 0);                                                               \
        }                                                                                               \
        asm volatile("" : "+v"(res));                                                                   \
        unsigned long long c1 = __builtin_readcyclecounter();                                           \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * 8 + wave] = Stamp{c0, c1};                         \
        out[blockIdx.x * 512 + threadIdx.x] = res;                                                      \
    }
PAIR(k_SS_0, BODY_S, BODY_S, 0)
PAIR(k_SS_2, BODY_S, BODY_S, 2)
PAIR(k_CC_0, BODY_C, BODY_C, 0)
PAIR(k_CC_2, BODY_C, BODY_C, 2)
PAIR(k_SC_0, BODY_S, BODY_C, 0)
PAIR(k_SC_2, BODY_S, BODY_C, 2)
PAIR(k_CS_0, BODY_C, BODY_S, 0)
PAIR(k_CS_2, BODY_C, BODY_S, 2)
PAIR(k_DS_0, BODY_D, BODY_S, 0)
PAIR(k_DS_2, BODY_D, BODY_S, 2)
PAIR(k_CD_0, BODY_C, BODY_D, 0)
PAIR(k_CD_2, BODY_C, BODY_D, 2)
PAIR(k_MS_0, BODY_M, BODY_S, 0)
PAIR(k_MS_2, BODY_M, BODY_S, 2)
PAIR(k_MC_0, BODY_M, BODY_C, 0)
PAIR(k_MC_2, BODY_M, BODY_C, 2)
PAIR(k_NS_0, BODY_N, BODY_S, 0)
PAIR(k_NS_2, BODY_N, BODY_S, 2)
PAIR(k_NC_0, BODY_N, BODY_C, 0)
PAIR(k_NC_2, BODY_N, BODY_C, 2)
PAIR(k_LS_0, BODY_L, BODY_S, 0)
PAIR(k_LS_2, BODY_L, BODY_S, 2)
PAIR(k_LC_0, BODY_L, BODY_C, 0)
PAIR(k_LC_2, BODY_L, BODY_C, 2)
PAIR(k_AC_0, BODY_A, BODY_C, 0)
PAIR(k_AC_2, BODY_A, BODY_C, 2)
PAIR(k_MM_0, BODY_M, BODY_M, 0)
PAIR(k_MM_2, BODY_M, BODY_M, 2)
PAIR(k_NN_0, BODY_N, BODY_N, 0)
PAIR(k_NN_2, BODY_N, BODY_N, 2)
PAIR(k_LL_0, BODY_L, BODY_L, 0)
PAIR(k_LL_2, BODY_L, BODY_L, 2)
PAIR(k_DD_0, BODY_D, BODY_D, 0)
PAIR(k_DD_2, BODY_D, BODY_D, 2)

typedef void (*kern_t)(uint32_t*, Stamp*, int, int);
static double median(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
// returns the median wave duration (cycles of the s_memtime counter = 100 MHz * ... use ratios only) of the X and of the Y waves
static void launch(kern_t k, int ix, int iy, double& dx, double& dy, double& ms) {
    const int blocks = 256 * 4;
    uint32_t* d_out; Stamp* d_st;
    (void)hipMalloc(&d_out, (size_t)blocks * 512 * 4);
    (void)hipMalloc(&d_st, sizeof(Stamp) * blocks * 8);
    (void)hipMemset(d_st, 0, sizeof(Stamp) * blocks * 8);
    for (int i = 0; i < 3; i++) k<<<blocks, 512>>>(d_out, d_st, ix, iy);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<<<blocks, 512>>>(d_out, d_st, ix, iy);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float f = 0; (void)hipEventElapsedTime(&f, e0, e1); ms = f;
    std::vector<Stamp> st(blocks * 8);
    (void)hipMemcpy(st.data(), d_st, sizeof(Stamp) * blocks * 8, hipMemcpyDeviceToHost);
    std::vector<double> vx, vy;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < 8; w++) (w < 4 ? vx : vy).push_back((double)(st[b * 8 + w].c1 - st[b * 8 + w].c0));
    dx = median(vx); dy = median(vy);
    (void)hipFree(d_out); (void)hipFree(d_st);
}
// three launches: X alone (Y waves idle), Y alone, both.  Work sized so that the two classes of the joint launch end together:
// first pass measures the solo rates, the joint launch gives each class iterations in inverse proportion.
static void run(const char* name, kern_t k) {
    const int base = 2000;
    double dx, dy, ms, d0;
    launch(k, base, 0, dx, d0, ms);
    const double x_alone = ms;
    launch(k, 0, base, d0, dy, ms);
    const double y_alone = ms;
    // joint: iterations such that solo times would be equal
    const int ix = base, iy = std::max(1, (int)(base * x_alone / y_alone + 0.5));
    double jx, jy;
    launch(k, ix, iy, jx, jy, ms);
    double sx, sy, m1, m2, t;
    launch(k, ix, 0, sx, t, m1);
    launch(k, 0, iy, t, sy, m2);
    printf("%-34s X alone %7.3f ms  Y alone %7.3f ms  (equal work)   both %7.3f ms = %4.2f x the sum, %4.2f x the longer one\n", name, m1, m2, ms,
           ms / (m1 + m2), ms / std::max(m1, m2));
    fflush(stdout);
}
#define RUN(x, y) run(#x " + " #y, k_##x##y##_0); run(#x " (priority 2) + " #y, k_##x##y##_2);
int main() {
    RUN(S, S)
    RUN(C, C)
    RUN(S, C)
    RUN(C, S)
    RUN(D, S)
    RUN(C, D)
    RUN(M, S)
    RUN(M, C)
    RUN(N, S)
    RUN(N, C)
    RUN(L, S)
    RUN(L, C)
    RUN(A, C)
    RUN(M, M)
    RUN(N, N)
    RUN(L, L)
    RUN(D, D)
    return 0;
}
