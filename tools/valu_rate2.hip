// valu_rate2.hip — second opcode sweep on MI355X: which integer / packed-16 / 24-bit ops issue at the full (2-cycle per wave64)
// VALU rate and could stand in for the half-rate rotates of Blake2s.  Measurement aid, not part of the product.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate2.hip -o tools/valu_rate2.bin ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t& x = a[i];
                const uint32_t y = a[(i + 1) & 7], z = a[(i + 2) & 7];
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 1) asm volatile("v_pk_add_u16 %0, %0, 0 op_sel:[1,0] op_sel_hi:[0,0]" : "+v"(x));
                if (OP == 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 3) asm volatile("v_pk_lshlrev_b16 %0, 3, %0" : "+v"(x));
                if (OP == 4) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 5) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 9) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(x) : "v"(y));
                if (OP == 12) asm volatile("v_bfe_u32 %0, %0, 5, 17" : "+v"(x));
                if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
                if (OP == 14) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 15) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 16) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 17) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 18) asm volatile("v_not_b32 %0, %0" : "+v"(x));
                if (OP == 19) asm volatile("v_bfrev_b32 %0, %0" : "+v"(x));
                if (OP == 20) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 21) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(y) : "vcc");
                if (OP == 22) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
                if (OP == 23) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(x));
                if (OP == 24) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(y));
                if (OP == 25) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 26) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 27) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x));
                if (OP == 28) asm volatile("v_pk_max_u16 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(x) : "v"(y));
                if (OP == 29) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 30) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x));
                if (OP == 31) asm volatile("v_xor_b32 %0, s2, %0" : "+v"(x));
                if (OP == 32) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 33) asm volatile("v_lshlrev_b16 %0, 3, %0" : "+v"(x));
                if (OP == 34) asm volatile("v_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 35) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 36) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 37) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 38) asm volatile("v_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 39) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                // pairs on live (non-decaying) data: a shift is full rate if the pair costs two full-rate slots, not three
                if (OP == 40) asm volatile("v_lshrrev_b32 %0, 7, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 41) asm volatile("v_lshlrev_b32 %0, 7, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 42) asm volatile("v_alignbit_b32 %0, %1, %1, 7\n\tv_xor_b32 %0, %0, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 43) asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 44) {
                    uint64_t p;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(x), "v"(y) : "vcc");
                    x = (uint32_t)p;
                }
                if (OP == 45) {
                    uint64_t p;
                    asm volatile("v_mad_u64_u32 %0, vcc, s2, %1, 0" : "=v"(p) : "v"(x) : "vcc");
                    x = (uint32_t)p;
                }
                // round 6: the 64-bit ops of the fold arithmetic (field.h m31_reduce64: v_lshl_add_u64, v_lshrrev_b64) and their 32-bit stand-ins
                if (OP == 49 || OP == 50 || OP == 51 || OP == 52) {
                    uint64_t q = ((uint64_t)y << 32) | x;
                    if (OP == 49) asm volatile("v_lshrrev_b64 %0, 31, %0" : "+v"(q));
                    if (OP == 50) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q) : "v"(q));
                    if (OP == 51) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q));
                    if (OP == 52) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(x), "v"(z) : "vcc");
                    x = (uint32_t)q;
                }
                if (OP == 53) asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(x));
                if (OP == 54) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(x));
                if (OP == 55) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 56) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(y));
                if (OP == 46) asm volatile("v_add_u32 %0, s2, %0" : "+v"(x));
                if (OP == 47) asm volatile("v_alignbit_b32 %0, %0, %1, s2" : "+v"(x) : "v"(y));
                if (OP == 48) asm volatile("v_min_u32 %0, s2, %0" : "+v"(x));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name) {
    uint32_t* d;
    int blocks = 256 * 8;
    (void)hipMalloc(&d, blocks * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 64;
    printf("%-44s %8.2f T lane-ops/s  (%.3f ms)\n", name, ops / (ms * 1e-3) / 1e12, ms);
    (void)hipFree(d);
}

int main() {
    run<0>("v_xor_b32");
    run<29>("v_add_u32");
    run<30>("v_add_u32 (32-bit literal)");
    run<31>("v_xor_b32 (sgpr operand)");
    run<1>("v_pk_add_u16 op_sel swap (= rotr 16)");
    run<2>("v_pk_add_u16");
    run<3>("v_pk_lshlrev_b16");
    run<4>("v_pk_mul_lo_u16");
    run<5>("v_pk_mad_u16");
    run<28>("v_pk_max_u16 op_sel");
    run<6>("v_mul_u32_u24");
    run<7>("v_mad_u32_u24");
    run<37>("v_mad_i32_i24");
    run<8>("v_mul_hi_u32_u24");
    run<20>("v_mad_u32_u16");
    run<9>("v_or3_b32");
    run<10>("v_and_or_b32");
    run<11>("v_lshl_or_b32");
    run<12>("v_bfe_u32");
    run<13>("v_cndmask_b32");
    run<14>("v_sad_u32");
    run<15>("v_dot4_u32_u8");
    run<16>("v_dot2_u32_u16");
    run<17>("v_xnor_b32");
    run<18>("v_not_b32");
    run<19>("v_bfrev_b32");
    run<21>("v_add_co_u32");
    run<22>("v_addc_co_u32");
    run<23>("v_lshrrev_b32");
    run<24>("v_mov_b32");
    run<25>("v_add_f32");
    run<26>("v_mul_f32");
    run<39>("v_fmac_f32");
    run<27>("v_cvt_f32_ubyte1");
    run<32>("v_add_u16");
    run<33>("v_lshlrev_b16");
    run<34>("v_mad_u16");
    run<38>("v_mul_lo_u16");
    run<35>("v_max_u32");
    run<36>("v_subrev_u32");
    run<43>("pair: v_xor + v_xor (per pair)");
    run<40>("pair: v_lshrrev_b32 + v_xor (per pair)");
    run<41>("pair: v_lshlrev_b32 + v_xor (per pair)");
    run<42>("pair: v_alignbit_b32 + v_xor (per pair)");
    run<44>("v_mad_u64_u32 (vgpr, vgpr)");
    run<45>("v_mad_u64_u32 (sgpr, vgpr)");
    run<46>("v_add_u32 (sgpr operand)");
    run<47>("v_alignbit_b32 (sgpr shift)");
    run<48>("v_min_u32 (sgpr operand)");
    run<49>("v_lshrrev_b64 (by 31)");
    run<50>("v_lshl_add_u64");
    run<51>("v_lshlrev_b64 (by 1)");
    run<52>("v_mad_u64_u32 (64-bit accumulate)");
    run<53>("v_lshlrev_b32");
    run<54>("v_and_b32 (literal)");
    run<55>("v_sub_u32");
    run<56>("v_lshl_add_u32");
    return 0;
}
