// Which bytes does a captured hipMemsetAsync node clear when the graph is replayed?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void dirty(unsigned char* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = 0xAB; }
int main() {
    const size_t sizes[] = {4, 360, 364, 4096, 8192 * 4, 32 * 2048 * 4};
    unsigned char* buf; CK(hipMalloc(&buf, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (size_t n : sizes) {
        for (int off = 0; off <= 4; off += 4) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            CK(hipMemsetAsync(buf + 256 + off, 0, n, s));
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(dirty, dim3(4096), dim3(256), 0, s, buf, 1 << 20);
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                std::vector<unsigned char> h(1 << 20);
                CK(hipMemcpy(h.data(), buf, 1 << 20, hipMemcpyDeviceToHost));
                size_t cleared = 0, stray = 0, first_bad = (size_t)-1;
                for (size_t i = 0; i < (1u << 20); ++i) {
                    bool in = i >= 256 + off && i < 256 + off + n;
                    if (in && h[i] == 0) ++cleared;
                    if (in && h[i] != 0 && first_bad == (size_t)-1) first_bad = i - 256 - off;
                    if (!in && h[i] != 0xAB) ++stray;
                }
                printf("n=%zu off=%d rep=%d cleared=%zu/%zu stray=%zu first_uncleared=%zd\n", n, off, rep, cleared, n, stray, (ssize_t)first_bad);
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
