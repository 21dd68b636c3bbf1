// LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU as a function of waves per CU, requests in flight per wave and the
// footprint the stream walks (L2-resident / Infinity-Cache-resident / HBM):  what bounds the operand supply of the GEMM loops.
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/ldsdma_bw tools/scratch/ldsdma_bw.hip ; ./tools/scratch/ldsdma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every wave streams `iters` x DEPTH KiB-pieces; piece p of wave w starts at ((base_of_wg + (w * iters * DEPTH + p)) * 1024) % footprint
template <int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(256) void stream_kernel(const float* src, size_t footprint_bytes, int iters, float* sink, int shared_by) {
    __shared__ f32x4 smem[4 * DEPTH * 64 + 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // `shared_by` workgroups walk the same addresses (operand panels shared behind one L2 / by neighbouring CUs)
    // distinct phase per group of workgroups (an odd number of KiB pieces apart), so that a small footprint is walked at
    // different places at any one time instead of every CU hitting the same lines together
    const size_t wg_base = (size_t)(blockIdx.x / shared_by) * (4 * iters * DEPTH + 977);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t piece = wg_base + ((size_t)wave_u * iters + it) * DEPTH + d;
            const size_t off = (piece * 1024) % footprint_bytes;
            if constexpr (TO_LDS) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + off / 4, 0, 1024, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (wave_u * DEPTH + d) * 64), 16, lane * 16, 0, 0, 0);
            } else {
                acc += *reinterpret_cast<const f32x4*>(src + off / 4 + lane * 4);
            }
        }
        if constexpr (TO_LDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += smem[(wave_u * DEPTH) * 64 + lane];
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int DEPTH, bool TO_LDS>
static void run(const float* src, size_t footprint, int wg_per_cu, int shared_by, float* sink, const char* what) {
    const int grid = 256 * wg_per_cu;
    const int iters = 4096 / DEPTH;                       // 4 MiB per wave
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_kernel<DEPTH, TO_LDS>), dim3(grid), dim3(256), 0, 0, src, footprint, iters, sink, shared_by);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)grid * 4 * iters * DEPTH * 1024;
    printf("%-8s depth %2d  waves/CU %2d  shared_by %2d  footprint %8.1f MiB : %7.2f TB/s  = %5.1f B/clk/CU at 2.1 GHz\n", what, DEPTH,
           wg_per_cu * 4, shared_by, footprint / 1048576.0, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    const size_t big = (size_t)4 << 30;
    float* src; float* sink;
    CHECK(hipMalloc(&src, big)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 0, big));
    const size_t fps[] = {(size_t)2 << 20, (size_t)8 << 20, (size_t)64 << 20, big};
    for (size_t fp : fps)
        for (int wg : {1, 2, 4}) {
            run<4, true>(src, fp, wg, 1, sink, "lds-dma");
            run<8, true>(src, fp, wg, 1, sink, "lds-dma");
            run<16, true>(src, fp, wg, 1, sink, "lds-dma");
            run<8, false>(src, fp, wg, 1, sink, "global");
            run<16, false>(src, fp, wg, 1, sink, "global");
        }
    // operand sharing as in the GEMMs: 4 / 8 workgroups walk the same panel at the same time (HBM-sized footprint)
    for (int sh : {2, 4, 8}) {
        run<8, true>(src, big, 2, sh, sink, "lds-dma");
        run<16, true>(src, big, 2, sh, sink, "lds-dma");
        run<16, false>(src, big, 2, sh, sink, "global");
    }
    return 0;
}
