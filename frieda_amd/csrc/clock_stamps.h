// clock_stamps.h — DIAGNOSTIC BUILDS ONLY (-DFRIEDA_CLOCK_STAMPS, tools/build_variant.sh clock -DFRIEDA_CLOCK_STAMPS): which clock does
// the chip hold inside the chip-filling product kernels?  Wave 0 of a workgroup reads the shader-clock counter (s_memtime) and the
// 100 MHz wall-clock counter (s_memrealtime) at its start and at its end into a buffer of its own that nothing else reads
// (MI355X_MICROARCH.md, "DVFS give-back" (6)); tools/wide_kernel_clock.py reads the buffers back after two seconds of proofs.  In the
// product build every macro below is empty: no product kernel carries a stamp.
#pragma once

#ifdef FRIEDA_CLOCK_STAMPS
#define FR_CLOCK_SLOTS 16384
#define FR_CLOCK_DECL(name) __device__ unsigned long long name[4 * FR_CLOCK_SLOTS];
// the stamp is tied to the data flow: `tie` (a VGPR value the kernel's work depends on) is redefined behind the first read
#define FR_CLOCK_BEGIN(tie)                                                                                                        \
    unsigned long long cs_c0_, cs_r0_;                                                                                             \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cs_c0_), "=s"(cs_r0_)::"memory"); \
    asm volatile("" : "+v"(tie) : "s"(cs_c0_));
// ... and the second read follows an asm that consumes `tie` (a value the kernel computed last)
#define FR_CLOCK_END(buf, tie)                                                                                                     \
    {                                                                                                                              \
        asm volatile("" ::"v"(tie));                                                                                               \
        unsigned long long cs_c1_, cs_r1_;                                                                                         \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cs_c1_), "=s"(cs_r1_)::"memory"); \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < FR_CLOCK_SLOTS) {                              \
            buf[4 * blockIdx.x + 0] = cs_c0_;                                                                                      \
            buf[4 * blockIdx.x + 1] = cs_r0_;                                                                                      \
            buf[4 * blockIdx.x + 2] = cs_c1_;                                                                                      \
            buf[4 * blockIdx.x + 3] = cs_r1_;                                                                                      \
        }                                                                                                                          \
    }
#define FR_CLOCK_READER(fn, buf)                                                                                                   \
    extern "C" int fn(unsigned long long* out, unsigned long n_words) {                                                            \
        const size_t bytes = sizeof(unsigned long long) * (n_words < 4 * FR_CLOCK_SLOTS ? n_words : 4 * FR_CLOCK_SLOTS);          \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(buf), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;                   \
    }
#else
#define FR_CLOCK_DECL(name)
#define FR_CLOCK_BEGIN(tie)
#define FR_CLOCK_END(buf, tie)
#define FR_CLOCK_READER(fn, buf)
#endif
