// Bit-exact device restatements of the two expf implementations the reference's postprocess reaches
// through torch-CPU (PyTorch 2.10, x86-64 with AVX-512; the reference itself pins neither).
//
// /root/reference/eval/orienmask_yolo_postprocess.py:126-139 calls .sigmoid() on STRIDED views of the
// [A,H,W,5+C] head tensor, so torch's TensorIterator picks the implementation per element:
//   * predict[..., 4] (objectness), pred_coord[..., 0/1] (tx, ty): inner stride 85 floats -> the scalar loop,
//     1 / (1 + std::exp(-x)) with glibc's expf (sysdeps/ieee754/flt-32/e_expf.c: the double-precision
//     2^(k/32) table algorithm of ARM's optimized routines);
//   * predict[..., 5:] (classes): rows of C contiguous floats -> the vectorised loop takes 2 x 16 lanes at a
//     time with Sleef's expf_u10 (FMA form), and the last C mod 32 elements of every row go through the scalar
//     loop again (ATen/native/cpu/Loops.h vectorized_loop).  For C = 80: classes 0..63 Sleef, 64..79 glibc.
// Both were pinned here by bit-comparison against torch on 4.5 M (glibc) / 9 M (Sleef) inputs and on the
// reference's own access patterns (0 mismatches; tools/probe_ref_math.py), and on the device by
// tests/test_hip_parity.py::test_ref_math_bit_exact.
//
// pred_coord[..., 2/3].exp() goes through MKL's vsExp (closed source, not correctly rounded: 1 % of inputs differ
// from the correctly rounded result by one ulp) -- box widths/heights are therefore matched to one ulp, not to the bit;
// the device uses the correctly rounded value (fp64 exp, one rounding).
//
// This file must be compiled with -ffp-contract=off (no fused multiply-adds except the explicit fma calls).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace om {

__device__ __constant__ const unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

// glibc expf (2.27+): exp(x) = 2^(k/32) * 2^(r/32), cubic in r, all in double, one rounding to float.
// tab: kExp2fTab or a copy of it in LDS (per-lane table indices make constant-memory reads serialise).
__device__ __forceinline__ float expf_glibc(float x, const unsigned long long* tab = kExp2fTab) {
    if (x != x) return x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();          // x > log(0x1p128)
    if (x < -0x1.9fe368p6f) return 0.0f;                     // x < log(0x1p-150)
    const double xd = (double)x;
    const double z0 = 0x1.71547652b82fep+5 * xd;             // x * 32 / ln2
    double kd = z0 + 0x1.8p+52;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= 0x1.8p+52;
    const double r = z0 - kd;
    const unsigned long long t = tab[ki & 31] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double z = 0x1.c6af84b912394p-20 * r + 0x1.ebfce50fac4f3p-13;
    const double r2 = r * r;
    double y = 0x1.62e42ff0c52d6p-6 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// Sleef 3.x xexpf (expf_u10), FMA build: what Vectorized<float>::exp() calls on AVX2/AVX-512.
__device__ __forceinline__ float expf_sleef(float d) {
    const float qf = rintf(d * 0x1.715476p+0f);              // R_LN2f
    const int q = (int)qf;
    float s = fmaf(qf, -0x1.62e4p-1f, d);                    // -L2Uf
    s = fmaf(qf, -0x1.7f7d1cp-20f, s);                       // -L2Lf
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    const int q1 = q >> 1;                                   // vldexp2: two exact power-of-two factors
    u = (u * __int_as_float((q1 + 127) << 23)) * __int_as_float((q - q1 + 127) << 23);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = __builtin_inff();
    return u;
}

__device__ __forceinline__ float sigmoid_scalar_ref(float x, const unsigned long long* tab = kExp2fTab) {
    return 1.0f / (1.0f + expf_glibc(-x, tab));
}
__device__ __forceinline__ float sigmoid_vector_ref(float x) { return 1.0f / (1.0f + expf_sleef(-x)); }

// Class c of a row of C class logits: the vectorised loop covers the first (C / 32) * 32 of them.
__device__ __forceinline__ float sigmoid_class_ref(float x, int c, int C, const unsigned long long* tab = kExp2fTab) {
    if (c < (C & ~31)) return sigmoid_vector_ref(x);
    return sigmoid_scalar_ref(x, tab);
}

// Correctly rounded expf for the box sizes (MKL's vsExp agrees with it on 99 % of inputs, else one ulp).
__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }

}  // namespace om
