// Attainable fp32 MFMA rate on this GPU: register-only v_mfma_f32_32x32x2_f32 loop, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 12345.678f) out[0] = s;
}
template <int CHAINS>
int run(int wg_per_cu, int iters, int ms_target) {
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)grid * 4 * iters * 8 * CHAINS * 4096.0;
        printf("chains=%d wg/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s  (= %.3f GHz x 256 CU x 256 flop/clk)\n", CHAINS, wg_per_cu, iters, ms,
               flops / ms / 1e9, flops / ms / 1e9 / (256 * 256) * 1e3 / 1e3);
    }
    return 0;
}
int main() {
    run<1>(3, 4000, 0); run<2>(2, 4000, 0); run<4>(1, 4000, 0); run<4>(2, 20000, 0);
    return 0;
}
