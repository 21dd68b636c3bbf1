// Stand-alone copy of the mask kernel's predicate loop, in the two forms discussed in orienmask_amd/csrc/post.hip
// (inside_bit) and profiles/r02_experiments.md section 6.  Not part of the library; built and run by run.py.
//   form 0 (compare):    (|Px - cx| < tx) && (|Py - cy| < ty) with floating-point compares: per detection 32 v_cmp_lt_f32_e64
//                        into SGPR pairs, s_and_b64, v_cndmask
//   form 1 (arithmetic): the same predicate on the bit patterns, everything in VGPRs
//   form 2 (vcc):        every compare as VOPC -> VCC -> v_cndmask, combined with VALU and
//   form 3 (branches):   nested divergent branches, v_cmp -> s_and_saveexec_b64
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

template <int FORM>
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
        if constexpr (FORM == 1) {
            const unsigned tx = threshold_bits(d.z), ty = threshold_bits(d.w);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned ax = __float_as_uint(Px[e] - d.x) & 0x7fffffffu, ay = __float_as_uint(Py[e] - d.y) & 0x7fffffffu;
                packed[e >> 2] |= ((sub_u32_opaque(ax, tx) & sub_u32_opaque(ay, ty)) >> 31) << ((e & 3) * 8);
            }
        } else if constexpr (FORM == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool inside = (fabsf(Px[e] - d.x) < d.z) && (fabsf(Py[e] - d.y) < d.w);
                packed[e >> 2] |= (inside ? 1u : 0u) << ((e & 3) * 8);
            }
        } else if constexpr (FORM == 2) {
            // every compare goes VOPC -> VCC -> v_cndmask at once (the opaque asm keeps the compiler from merging lane masks)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                unsigned ix = (fabsf(Px[e] - d.x) < d.z) ? 1u : 0u;
                asm("" : "+v"(ix));
                unsigned iy = (fabsf(Py[e] - d.y) < d.w) ? 1u : 0u;
                asm("" : "+v"(iy));
                packed[e >> 2] |= (ix & iy) << ((e & 3) * 8);
            }
        } else {
            // real divergent branches: v_cmp -> s_and_saveexec_b64 (the asm cannot be if-converted)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (fabsf(Px[e] - d.x) < d.z) {
                    asm volatile("s_nop 0");
                    if (fabsf(Py[e] - d.y) < d.w) {
                        asm volatile("s_nop 0");
                        packed[e >> 2] |= 1u << ((e & 3) * 8);
                    }
                }
            }
        }
        uint4 o;
        o.x = packed[0]; o.y = packed[1]; o.z = packed[2]; o.w = packed[3];
        out[(size_t)i * gridDim.x * 256 + t] = o;
    }
}

}  // namespace

// neighbours that do nothing but issue matrix instructions on registers (no memory traffic, 64 VGPRs at most): which matrix
// instruction of a co-resident wave disturbs the compare form?
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void mfma_neighbour(float* sink, int iters) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4v acc4 = {0.f, 0.f, 0.f, 0.f};
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    f16x8 ah, bh;
    bf16x8 ab, bb;
#pragma unroll
    for (int k = 0; k < 8; ++k) { ah[k] = (_Float16)(seed + k); bh[k] = (_Float16)(1.f / (1 + k)); ab[k] = (__bf16)(seed + k); bb[k] = (__bf16)(1.f / (1 + k)); }
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if constexpr (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.5f, acc, 0, 0, 0);
        if constexpr (KIND == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
        if constexpr (KIND == 3) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc4, 0, 0, 0);
    }
    float t = acc4[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[r];
    if (t == 12345.f) sink[0] = t;
}

extern "C" int neighbour_launch(int kind, float* sink, int blocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(mfma_neighbour<0>, dim3(blocks), dim3(256), 0, s, sink, iters);
    else if (kind == 1) hipLaunchKernelGGL(mfma_neighbour<1>, dim3(blocks), dim3(256), 0, s, sink, iters);
    else if (kind == 2) hipLaunchKernelGGL(mfma_neighbour<2>, dim3(blocks), dim3(256), 0, s, sink, iters);
    else hipLaunchKernelGGL(mfma_neighbour<3>, dim3(blocks), dim3(256), 0, s, sink, iters);
    return (int)hipGetLastError();
}

extern "C" int probe_launch(int form, const float* seed, const float4* dets, int n, uint4* out, int blocks, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (form == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else hipLaunchKernelGGL(probe_kernel<3>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    return (int)hipGetLastError();
}
