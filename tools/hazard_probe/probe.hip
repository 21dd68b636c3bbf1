// Stand-alone copy of the mask kernel's predicate loop, in the two forms discussed in orienmask_amd/csrc/post.hip
// (inside_bit) and profiles/r02_experiments.md section 6.  Not part of the library; built and run by run.py.
//   probe_compare:    (|Px - cx| < tx) && (|Py - cy| < ty) with floating-point compares: per detection 32 v_cmp_lt_f32_e64
//                     into SGPR pairs, s_and_b64, v_cndmask
//   probe_arithmetic: the same predicate on the bit patterns, everything in VGPRs
// Both write one uint4 (16 predicate bytes) per thread and detection; run.py compares the outputs of repeated launches with
// and without another stream's fp16 convolution resident.
#include <hip/hip_runtime.h>

namespace {

__device__ __forceinline__ unsigned threshold_bits(float t) {
    const unsigned u = __float_as_uint(t);
    return u <= 0x7f800000u ? u : 0u;
}
__device__ __forceinline__ unsigned sub_u32_opaque(unsigned a, unsigned b) {
    unsigned r;
    asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool ARITH>
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ seed, const float4* __restrict__ dets, int n, uint4* out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float Px[16], Py[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        Px[e] = seed[(t * 16 + e) & 65535] * 8.f;
        Py[e] = seed[(t * 16 + e + 7777) & 65535] * 8.f;
    }
    __shared__ float4 s_det[256];
    s_det[threadIdx.x] = dets[threadIdx.x % n];
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const float4 d = s_det[i];
        unsigned packed[4] = {0, 0, 0, 0};
        if constexpr (ARITH) {
            const unsigned tx = threshold_bits(d.z), ty = threshold_bits(d.w);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned ax = __float_as_uint(Px[e] - d.x) & 0x7fffffffu, ay = __float_as_uint(Py[e] - d.y) & 0x7fffffffu;
                packed[e >> 2] |= ((sub_u32_opaque(ax, tx) & sub_u32_opaque(ay, ty)) >> 31) << ((e & 3) * 8);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool inside = (fabsf(Px[e] - d.x) < d.z) && (fabsf(Py[e] - d.y) < d.w);
                packed[e >> 2] |= (inside ? 1u : 0u) << ((e & 3) * 8);
            }
        }
        uint4 o;
        o.x = packed[0]; o.y = packed[1]; o.z = packed[2]; o.w = packed[3];
        out[(size_t)i * gridDim.x * 256 + t] = o;
    }
}

}  // namespace

extern "C" int probe_launch(int arithmetic, const float* seed, const float4* dets, int n, uint4* out, int blocks, void* stream) {
    if (arithmetic) hipLaunchKernelGGL(probe_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed, dets, n, out);
    else hipLaunchKernelGGL(probe_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed, dets, n, out);
    return (int)hipGetLastError();
}
