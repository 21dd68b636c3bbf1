// LDS-DMA throughput of the GEMM operand access pattern: a workgroup streams panels of 64 rows x S bytes, one k-step (128 bytes
// of every row) at a time, as the Winograd / implicit GEMM loaders do (S = 4 C bytes per row).  Against the same bytes laid out
// k-step-major (the 64 x 128-byte rows of a k-step contiguous: S = 128).  L2-resident, Infinity-Cache-sized and HBM footprints.
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/ldsdma_stride tools/scratch/ldsdma_stride.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// panel p of a workgroup: bytes [p * 64 * S, (p + 1) * 64 * S) of its region (mod footprint); k-step c reads bytes [c * 128, c * 128 + 128)
// of each of the 64 rows; wave w issues pieces 2 w and 2 w + 1 (8 rows each); KS k-steps are requested before each wait.
template <int KS>
__global__ __launch_bounds__(256) void panel_kernel(const char* src, size_t footprint, int S, int panels, float* sink, int shared_by,
                                                    size_t group_stride) {
    __shared__ f32x4 smem[KS * 8 * 64 + 4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int kc = S / 128;
    const size_t base = ((size_t)(blockIdx.x / shared_by) * group_stride) % footprint;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int lane_off = (lane >> 3) * S + (lane & 7) * 16;
    for (int p = 0; p < panels; ++p) {
        const size_t pb = (base + (size_t)p * 64 * S) % footprint;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + pb, 0, 64 * S, 0x00020000);
        for (int c0 = 0; c0 < kc; c0 += KS) {
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (c0 + k < kc) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + ((k * 8) + wave * 2 + j) * 64), 16,
                                                                 lane_off + (wave * 2 + j) * 8 * S, (c0 + k) * 128, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += smem[wave * 2 * 64 + lane];
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int KS>
static void run(const char* src, size_t footprint, int S, int wg_per_cu, int shared_by, float* sink) {
    const int grid = 256 * wg_per_cu;
    const size_t per_wg = (size_t)16 << 20;                  // 16 MiB per workgroup
    const int panels = (int)(per_wg / (64 * (size_t)S));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((panel_kernel<KS>), dim3(grid), dim3(256), 0, 0, src, footprint, S, panels, sink, shared_by,
                           per_wg + 977 * 1024);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    const double bytes = (double)grid * panels * 64.0 * S;
    printf("row stride %5d B  k-steps in flight %d  waves/CU %2d  shared_by %d  footprint %7.1f MiB : %6.2f TB/s = %5.1f B/clk/CU\n", S, KS,
           wg_per_cu * 4, shared_by, footprint / 1048576.0, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    const size_t big = (size_t)4 << 30;
    char* src; float* sink;
    CHECK(hipMalloc(&src, big + (64 << 20))); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 0, big + (64 << 20)));
    for (size_t fp : {(size_t)2 << 20, (size_t)64 << 20, big})
        for (int S : {128, 256, 512, 1024, 2048, 4096})
            for (int wg : {2, 4}) {
                run<3>(src, fp, S, wg, 1, sink);
            }
    // operand sharing: 4 workgroups stream the same panels at the same time (the N tiles of a V panel; every M panel reads U)
    for (int S : {128, 512, 2048}) {
        run<3>(src, big, S, 2, 4, sink);
        run<3>(src, (size_t)64 << 20, S, 2, 4, sink);
        run<3>(src, (size_t)64 << 20, S, 2, 32, sink);
    }
    return 0;
}
