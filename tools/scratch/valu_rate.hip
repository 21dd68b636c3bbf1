// Issue rate of a straight-line VALU stream from ONE wave per SIMD (the producers of conv_wino14.hip): cycles per instruction by
// s_memtime around 1024 instructions, for 4-byte and 8-byte encodings and for the conversion / mixed-precision ops of the hi/lo split.
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/valu_rate tools/scratch/valu_rate.hip && tools/scratch/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define BODY8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define KERNEL(NAME, ASM8)                                                                                              \
    __global__ void NAME(unsigned long long* out, float seed) {                                                         \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;  \
        float b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3, b4 = seed + 4, b5 = seed + 5, b6 = seed + 6, b7 = seed + 7;  \
        unsigned long long t0, t1;                                                                                      \
        __syncthreads();                                                                                                \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                       \
        asm volatile(".rept 128\n" ASM8 ".endr\n"                                                                       \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2),    \
                       "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));                                               \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                              \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 12345.f) out[0] = 0;      \
    }

// independent streams over 8 register pairs
KERNEL(k_fmac_e32, "v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %9, %10\n v_fmac_f32_e32 %2, %10, %11\n v_fmac_f32_e32 %3, %11, %12\n"
                   "v_fmac_f32_e32 %4, %12, %13\n v_fmac_f32_e32 %5, %13, %14\n v_fmac_f32_e32 %6, %14, %15\n v_fmac_f32_e32 %7, %15, %8\n")
KERNEL(k_fma_e64, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %9, %10, %1\n v_fma_f32 %2, %10, %11, %2\n v_fma_f32 %3, %11, %12, %3\n"
                  "v_fma_f32 %4, %12, %13, %4\n v_fma_f32 %5, %13, %14, %5\n v_fma_f32 %6, %14, %15, %6\n v_fma_f32 %7, %15, %8, %7\n")
KERNEL(k_fma_dep, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %0, %9, %10, %0\n v_fma_f32 %0, %10, %11, %0\n v_fma_f32 %0, %11, %12, %0\n"
                  "v_fma_f32 %0, %12, %13, %0\n v_fma_f32 %0, %13, %14, %0\n v_fma_f32 %0, %14, %15, %0\n v_fma_f32 %0, %15, %8, %0\n")
KERNEL(k_cvt_pk, "v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %9, %10\n v_cvt_pk_f16_f32 %2, %10, %11\n v_cvt_pk_f16_f32 %3, %11, %12\n"
                 "v_cvt_pk_f16_f32 %4, %12, %13\n v_cvt_pk_f16_f32 %5, %13, %14\n v_cvt_pk_f16_f32 %6, %14, %15\n v_cvt_pk_f16_f32 %7, %15, %8\n")
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %8, -1.0, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %9, -1.0, %10 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                  "v_fma_mix_f32 %2, %10, -1.0, %11 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %11, -1.0, %12 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                  "v_fma_mix_f32 %4, %12, -1.0, %13 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %13, -1.0, %14 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                  "v_fma_mix_f32 %6, %14, -1.0, %15 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %15, -1.0, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
KERNEL(k_sub_e32, "v_sub_f32_e32 %0, %8, %9\n v_sub_f32_e32 %1, %9, %10\n v_sub_f32_e32 %2, %10, %11\n v_sub_f32_e32 %3, %11, %12\n"
                  "v_sub_f32_e32 %4, %12, %13\n v_sub_f32_e32 %5, %13, %14\n v_sub_f32_e32 %6, %14, %15\n v_sub_f32_e32 %7, %15, %8\n")

// packed fp32 on register pairs
#define KERNEL_PK(NAME, ASM4)                                                                                           \
    __global__ void NAME(unsigned long long* out, float seed) {                                                         \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                           \
        f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b0 = a0, b1 = a1, b2 = a2, b3 = a3;          \
        unsigned long long t0, t1;                                                                                      \
        __syncthreads();                                                                                                \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                       \
        asm volatile(".rept 256\n" ASM4 ".endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));  \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                              \
        f2 s = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;                                                                   \
        if (s[0] + s[1] == 12345.f) out[0] = 0;                                                                         \
    }
KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %5, %6, %1\n v_pk_fma_f32 %2, %6, %7, %2\n v_pk_fma_f32 %3, %7, %4, %3\n")
KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %4, %5\n v_pk_add_f32 %1, %5, %6\n v_pk_add_f32 %2, %6, %7\n v_pk_add_f32 %3, %7, %4\n")

template <typename K>
void run(const char* name, K k, int threads, unsigned long long* d, int bytes_per_instr) {
    unsigned long long h[256 * 16];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 1.0f);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 16 + w]; ++n; }
    printf("%-12s %2d waves/CU  %6.2f cycles / instruction  (%d-byte encoding)\n", name, threads / 64, s / n / 1024.0, bytes_per_instr);
}

int main() {
    unsigned long long* d;
    CHECK(hipMalloc(&d, 256 * 16 * 8));
    for (int threads : {64, 256, 512, 768}) {
        run("fmac_e32", k_fmac_e32, threads, d, 4);
        run("sub_e32", k_sub_e32, threads, d, 4);
        run("fma_e64", k_fma_e64, threads, d, 8);
        run("fma_e64 dep", k_fma_dep, threads, d, 8);
        run("pk_fma", k_pk_fma, threads, d, 8);
        run("pk_add", k_pk_add, threads, d, 8);
        run("cvt_pk_f16", k_cvt_pk, threads, d, 8);
        run("fma_mix", k_fma_mix, threads, d, 8);
    }
    return 0;
}
