// Does a captured linear chain of kernel nodes keep kernel->kernel ordering on this ROCm?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void slow_writer(int* flag, long long cycles) {
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < cycles) {}
        *flag = 1;
    }
}
__global__ void reader(const int* flag, int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = *(volatile const int*)flag;
}
int main() {
    int *flag, *out; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 1024 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    int h[1024];
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemsetAsync(flag, 0, 4, s)); CK(hipMemsetAsync(out, 0xff, 4096, s)); CK(hipStreamSynchronize(s));
        if (mode == 0) {
            hipLaunchKernelGGL(slow_writer, dim3(1024), dim3(256), 0, s, flag, 100000000LL / 10);  // 100 MHz clock -> 100 ms
            hipLaunchKernelGGL(reader, dim3(1024), dim3(64), 0, s, flag, out);
        } else {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            hipLaunchKernelGGL(slow_writer, dim3(1024), dim3(256), 0, s, flag, 100000000LL / 10);
            hipLaunchKernelGGL(reader, dim3(1024), dim3(64), 0, s, flag, out);
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s));
        }
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost));
        int ones = 0; for (int i = 0; i < 1024; ++i) ones += h[i] == 1;
        printf("%s: reader blocks that saw the flag: %d / 1024\n", mode ? "graph" : "eager", ones);
    }
    return 0;
}
