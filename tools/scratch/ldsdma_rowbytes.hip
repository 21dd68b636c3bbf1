// Does the CU's request path care whether an operand row arrives 64 or 128 bytes per k-step?  (A 16-channel fp32 chunk of a pixel is
// half a 128-byte line: the k-loops of conv_igemm_split.hip and the producers of conv_wino14.hip fetch half lines.)
// A workgroup streams 16 KiB per step -- R rows x WB bytes, rows S bytes apart, consecutive steps WB further along the rows --
// by LDS-DMA with three steps in flight; every workgroup its own rows.
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/ldsdma_rowbytes tools/scratch/ldsdma_rowbytes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WB>
__global__ __launch_bounds__(256) void stream_kernel(const char* src, size_t footprint, int S, int steps, float* sink) {
    constexpr int ROWS = 16384 / WB;                  // rows per step
    constexpr int LPR = WB / 16;                      // lanes per row
    __shared__ f32x4 smem[3 * 1024 + 4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * ROWS * S) % footprint;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + base, 0, ROWS * S, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ksteps_per_row = S / WB;
    auto issue = [&](int step, int buf) {
        const int kc = step % ksteps_per_row;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                  // 16 pieces of 1 KiB per step, four per wave
            const int piece = wave * 4 + j;
            const int row = piece * (64 / LPR) + lane / LPR;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * 1024 + piece * 64), 16, row * S + (lane % LPR) * 16, kc * WB, 0, 0);
        }
    };
    issue(0, 0); issue(1, 1);
    for (int s = 0; s < steps; ++s) {
        issue(s + 2, (s + 2) % 3);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        acc += smem[(s % 3) * 1024 + threadIdx.x];
        __syncthreads();
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int WB>
static void run(const char* src, size_t footprint, int S, int wg_per_cu, float* sink) {
    const int grid = 256 * wg_per_cu, steps = 2048;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_kernel<WB>), dim3(grid), dim3(256), 0, 0, src, footprint, S, steps, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    const double bytes = (double)grid * steps * 16384.0;
    printf("%3d B of a row per step  row stride %5d B  workgroups/CU %d  footprint %7.1f MiB : %6.2f TB/s = %5.1f B/clk/CU\n", WB, S, wg_per_cu,
           footprint / 1048576.0, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    const size_t big = (size_t)2 << 30;
    char* src; float* sink;
    CHECK(hipMalloc(&src, big + (64 << 20))); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 0, big + (64 << 20)));
    for (size_t fp : {(size_t)16 << 20, (size_t)128 << 20, big})
        for (int S : {512, 2048})
            for (int wg : {2, 3}) {
                run<64>(src, fp, S, wg, sink);
                run<128>(src, fp, S, wg, sink);
                run<256>(src, fp, S, wg, sink);
            }
    return 0;
}
