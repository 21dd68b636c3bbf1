// Stand-alone copy of the mask kernel's predicate loop, in the two forms discussed in orienmask_amd/csrc/post.hip
// (inside_bit) and profiles/r02_experiments.md section 6.  Not part of the library; built and run by run.py.
//   form 0 (compare):    (|Px - cx| < tx) && (|Py - cy| < ty) with floating-point compares: per detection 32 v_cmp_lt_f32_e64
//                        into SGPR pairs, s_and_b64, v_cndmask
//   form 1 (arithmetic): the same predicate on the bit patterns, everything in VGPRs
//   form 2 (vcc):        every compare as VOPC -> VCC -> v_cndmask, combined with VALU and
//   form 3 (branches):   nested divergent branches, v_cmp -> s_and_saveexec_b64
//   forms 4, 5, 6:       the compare form in inline asm with 0 / 4+1 / 15+1 wait states (s_nop) between the VALU compares and
//                        the s_and_b64 that reads their SGPR pairs
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
        } else if constexpr (FORM == 8) {
            // the compare form exactly as form 0, but the differences are formed by scalar v_sub_f32 (opaque asm: the compiler cannot
            // pack them into v_pk_add_f32): do the failures need the packed subtractions in front of the compares?
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float dx, dy;
                asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(Px[e]), "v"(d.x));
                asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(Py[e]), "v"(d.y));
                const bool inside = (fabsf(dx) < d.z) && (fabsf(dy) < d.w);
                packed[e >> 2] |= (inside ? 1u : 0u) << ((e & 3) * 8);
            }
        } else if constexpr (FORM == 9) {
            // hand-written: PACKED subtractions (v_pk_add_f32 with neg modifiers, as the compiler emits them) straight in front of
            // v_cmp_lt_f32_e64 into SGPR pairs, s_and_b64, v_cndmask -- no wait states anywhere
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                float2 ax, ay;
                const float2 px = {Px[e], Px[e + 1]}, py = {Py[e], Py[e + 1]};
                const float2 cx = {d.x, d.x}, cy = {d.y, d.y};
                unsigned long long m0, m1, m2, m3, b0, b1;
                unsigned bit0, bit1;
                asm volatile("v_pk_add_f32 %0, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %1, %3, %5 neg_lo:[0,1] neg_hi:[0,1]"
                             : "=&v"(ax), "=&v"(ay) : "v"(px), "v"(py), "v"(cx), "v"(cy));
                asm volatile("v_cmp_lt_f32_e64 %0, |%8|, %12\n\t"
                             "v_cmp_lt_f32_e64 %1, |%9|, %12\n\t"
                             "v_cmp_lt_f32_e64 %2, |%10|, %13\n\t"
                             "v_cmp_lt_f32_e64 %3, |%11|, %13\n\t"
                             "s_and_b64 %4, %0, %2\n\t"
                             "s_and_b64 %5, %1, %3\n\t"
                             "v_cndmask_b32_e64 %6, 0, 1, %4\n\t"
                             "v_cndmask_b32_e64 %7, 0, 1, %5"
                             : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(b0), "=&s"(b1), "=&v"(bit0), "=&v"(bit1)
                             : "v"(ax.x), "v"(ax.y), "v"(ay.x), "v"(ay.y), "v"(d.z), "v"(d.w) : "scc");
                packed[e >> 2] |= bit0 << ((e & 3) * 8);
                packed[(e + 1) >> 2] |= bit1 << (((e + 1) & 3) * 8);
            }
        } else if constexpr (FORM == 10 || FORM == 11) {
            // the compiler's PACKED subtractions verbatim (one register pair holds (cx, cy); op_sel_hi:[1,0] broadcasts cx, op_sel:[0,1]
            // broadcasts cy), in asm; behind them form 10 evaluates the predicate on bit patterns in VGPRs (no compare at all), form 11
            // with the compiler's floating-point compares
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                float2 ax, ay;
                const float2 px = {Px[e], Px[e + 1]}, py = {Py[e], Py[e + 1]};
                const float2 cxy = {d.x, d.y};
                asm volatile("v_pk_add_f32 %0, %2, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             "v_pk_add_f32 %1, %3, %4 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
                             : "=&v"(ax), "=&v"(ay) : "v"(px), "v"(py), "v"(cxy));
                const float dxs[2] = {ax.x, ax.y}, dys[2] = {ay.x, ay.y};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    unsigned bit;
                    if constexpr (FORM == 10) {
                        const unsigned tx = threshold_bits(d.z), ty = threshold_bits(d.w);
                        const unsigned bx = __float_as_uint(dxs[k]) & 0x7fffffffu, by = __float_as_uint(dys[k]) & 0x7fffffffu;
                        bit = (sub_u32_opaque(bx, tx) & sub_u32_opaque(by, ty)) >> 31;
                    } else {
                        bit = ((fabsf(dxs[k]) < d.z) && (fabsf(dys[k]) < d.w)) ? 1u : 0u;
                    }
                    packed[(e + k) >> 2] |= bit << (((e + k) & 3) * 8);
                }
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
        } else if constexpr (FORM == 7) {
            // as the compiler schedules it: the lane masks are combined into VCC by SALU, further VALU compares (writing other SGPR
            // pairs) are issued, THEN v_cndmask_b32_e32 reads VCC
            unsigned long long mx[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ax = Px[e] - d.x;
                asm volatile("v_cmp_lt_f32_e64 %0, |%1|, %2" : "=s"(mx[e]) : "v"(ax), "v"(d.z));
            }
            unsigned long long my;
            {
                const float ay0 = Py[0] - d.y;
                asm volatile("v_cmp_lt_f32_e64 %0, |%1|, %2" : "=s"(my) : "v"(ay0), "v"(d.w));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ay1 = Py[(e + 1) & 15] - d.y;
                unsigned long long mynext;
                unsigned bit;
                const unsigned one = 1u << ((e & 3) * 8);
                asm volatile("s_and_b64 vcc, %2, %3\n\tv_cmp_lt_f32_e64 %1, |%4|, %5\n\tv_cndmask_b32_e32 %0, 0, %6, vcc"
                             : "=v"(bit), "=&s"(mynext) : "s"(mx[e]), "s"(my), "v"(ay1), "v"(d.w), "v"(one) : "vcc", "scc");
                my = mynext;
                packed[e >> 2] |= bit;
            }
        } else if constexpr (FORM >= 4) {
            // the compare form written by hand with NOPS wait states between the VALU compares and the SALU that reads their
            // SGPR pairs: 16 v_cmp_e64 into 16 SGPR pairs back to back (x), s_nop, then per element v_cmp (y) + s_and_b64 + v_cndmask
            constexpr int NOPS = FORM == 4 ? 0 : FORM == 5 ? 4 : 15;
            unsigned long long mx[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ax = fabsf(Px[e] - d.x);
                asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mx[e]) : "v"(ax), "v"(d.z));
            }
            asm volatile("s_nop %0" ::"i"(NOPS));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ay = fabsf(Py[e] - d.y);
                unsigned long long my, both;
                unsigned bit;
                asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(my) : "v"(ay), "v"(d.w));
                asm volatile("s_nop %0" ::"i"(NOPS));
                asm volatile("s_and_b64 %0, %1, %2" : "=s"(both) : "s"(mx[e]), "s"(my) : "scc");
                asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(bit) : "s"(both));
                packed[e >> 2] |= bit << ((e & 3) * 8);
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
    else if (form == 3) hipLaunchKernelGGL(probe_kernel<3>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 4) hipLaunchKernelGGL(probe_kernel<4>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 5) hipLaunchKernelGGL(probe_kernel<5>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 6) hipLaunchKernelGGL(probe_kernel<6>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 7) hipLaunchKernelGGL(probe_kernel<7>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 8) hipLaunchKernelGGL(probe_kernel<8>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 9) hipLaunchKernelGGL(probe_kernel<9>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else if (form == 10) hipLaunchKernelGGL(probe_kernel<10>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    else hipLaunchKernelGGL(probe_kernel<11>, dim3(blocks), dim3(256), 0, s, seed, dets, n, out);
    return (int)hipGetLastError();
}
