// valu_rate.hip — microbenchmark: sustained per-chip rate of the integer VALU ops Blake2s and M31 arithmetic are made
// of, next to fp32 FMA.  Measurement aid, not part of the product.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate.bin ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters) {
    uint32_t a[8];
    float f[8];
    double pk[8];
    for (int i = 0; i < 8; i++) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
        f[i] = (float)a[i] * 1e-9f;
        pk[i] = (double)a[i];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t& x = a[i];
                const uint32_t y = a[(i + 1) & 7], z = a[(i + 2) & 7];
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));
                if (OP == 3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 6) {
                    uint64_t p;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(x), "v"(y) : "vcc");
                    x = (uint32_t)p;
                }
                if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 8) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 9) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 10) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<uint64_t*>(&pk[i])) : "v"(pk[(i + 1) & 7]), "v"(pk[(i + 2) & 7]));
                if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(y));
                if (OP == 13) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 14) asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 15) asm volatile("v_xor_b32_dpp %0, %1, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 16) asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(x));
                if (OP == 17) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 18) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 19) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (OP == 20) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
                if (OP == 21) asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
                if (OP == 22) asm volatile("v_alignbyte_b32 %0, %0, %0, 1" : "+v"(x));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)f[i] + (uint32_t)pk[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, double ops_per_inner) {
    uint32_t* d;
    int blocks = 256 * 8;
    (void)hipMalloc(&d, blocks * 256 * 4);
    int iters = 4000;
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
    double ops = (double)blocks * 256 * iters * 64 * ops_per_inner;
    printf("%-28s %8.2f T lane-ops/s  (%.3f ms)\n", name, ops / (ms * 1e-3) / 1e12, ms);
    (void)hipFree(d);
}

int main() {
    run<0>("v_xor_b32", 1);
    run<1>("v_add_u32", 1);
    run<2>("v_alignbit_b32", 1);
    run<3>("v_add3_u32", 1);
    run<4>("v_fma_f32", 1);
    run<5>("v_min_u32", 1);
    run<6>("v_mad_u64_u32", 1);
    run<7>("v_mul_lo_u32", 1);
    run<8>("v_mul_hi_u32", 1);
    run<9>("v_perm_b32", 1);
    run<10>("v_xad_u32", 1);
    run<11>("v_pk_fma_f32 (2 lanes-ops)", 2);
    run<12>("v_lshl_add_u32", 1);
    run<13>("v_bfi_b32", 1);
    run<14>("v_xor_b32_sdwa", 1);
    run<15>("v_xor_b32_dpp quad_perm", 1);
    run<16>("v_lshlrev_b32", 1);
    run<17>("v_or_b32", 1);
    run<18>("v_sub_u32", 1);
    run<19>("v_and_b32", 1);
    run<20>("v_mov_b32_dpp quad_perm", 1);
    run<21>("v_add_u32_dpp quad_perm", 1);
    run<22>("v_alignbyte_b32", 1);
    return 0;
}
