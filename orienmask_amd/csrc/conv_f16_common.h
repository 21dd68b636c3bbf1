// Shared pieces of the fp16-activation convolution kernels (conv_igemm_f16.hip, conv3x3_f16.hip).
#pragma once
#include "om_common.h"

namespace om {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct IgemmHParams {
    const _Float16* in;
    const _Float16* w;
    const float* scale;
    const float* shift;
    const _Float16* res;
    void* out;            // fp16 (out_f32 = 0) or fp32 (out_f32 = 1)
    int* ticket;
    int H, W, cin, in_pix_stride;
    int Ho, Wo, HoWo, cout;
    int ks, stride, pad;
    int M, kc, ksteps, taps;      // kc = cin / 32, ksteps = taps * kc
    int n_tiles, total_tiles;
    int leaky, res_pix_stride, out_pix_stride, out_mode, up, out_f32;
    int vec_io;
    int total_in_pixels;
    int w_bytes;
    // gathered input of a 1x1 layer (conv_igemm_f16_kernel<..., GATHER>; as conv_igemm_split.hip's): the cin channels are the
    // concatenation of nseg tensors, segment g holding the 32-channel chunks [seg_end[g-1], seg_end[g]) and stored at
    // 1 / 2^seg_shift[g] of this layer's resolution -- the nearest-neighbour up-sampling of the reference's routes and skips
    // (orienmask_yolo_fpnplus.py:78-86) happens in the operand addresses instead of in replicated stores
    int nseg, nimg;
    const _Float16* seg_ptr[4];
    int seg_stride[4], seg_shift[4], seg_end[4];      // pixel stride in halfs
};

// Epilogue of a finished BM x BN tile whose accumulators are in the transposed 32x32 MFMA layout
// (pixel = lane & 31, channel = 8*(r>>2) + 4*(lane>>5) + (r&3)): one wave-row (WM pixels x BN channels) at a time through
// LDS in fp32 (16-byte chunk index swizzled with m & 7), then 8 channels per thread: scale/shift, LeakyReLU, residual,
// one 16-byte fp16 store (fp32 for the head tensors; NCHW fp32 for the orientation head).  `smem` must hold
// WM * BN / 4 f32x4 and be free of live operands; all NT threads of the workgroup call it.
//
// FAST (the launchers: fp16 NHWC output, no residual, 16-byte aligned view, cout == cout_pad): row sweeps without any load.  With
// the residual's conditional loads in the sweep loop the compiler waits for vmcnt(0) at the top of every sweep, i.e. for the
// PREVIOUS sweep's stores (one in-order counter for loads and stores): conv_igemm_split.hip's epilogue, DESIGN.md 3.5.
// FAST = 2: the same with a residual (fp16 NHWC, 16-byte aligned): its rows are requested RG sweeps at a time BEFORE those
// sweeps' stores, so a pass waits for the previous stores WM / RP / RG times instead of once per sweep.
template <int BM, int BN, int WM, int WN, int FAST = 0, int NT = 256>
__device__ __forceinline__ void f16_epilogue(const IgemmHParams& p, f32x4* smem, const f32x16 (&acc)[WM / 32][WN / 32],
                                             int m0, int n0, int tid, int wm, int wn, int fi, int fk) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int CH8 = BN / 8;                   // 8-channel chunks per C-tile row
    constexpr int RP = NT / CH8;                  // C-tile rows per epilogue sweep
    constexpr int CH = BN / 4;                    // f32x4 chunks per C-tile row
    f32x4* sC = smem;
    const int n8 = tid % CH8, r0 = tid / CH8;
    const int n = n0 + n8 * 8;
    const int nvalid = p.cout - n;
    const bool vec = p.vec_io && nvalid >= 8;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = p.scale[n + k]; sh[k] = p.shift[n + k]; }   // padded to cout_pad
#pragma unroll 1
    for (int pass = 0; pass < BM / WM; ++pass) {
        // phase 1: accumulators -> LDS C tile [m][n] fp32, 16-byte chunk index swizzled with m & 7.
        // Transposed 32x32 D layout: pixel = lane & 31, channel = 8*(r>>2) + 4*(lane>>5) + (r&3).
        if (wm == pass) {
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int ml = a * 32 + fi;
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n4 = (wn * WN + b * 32) / 4 + 2 * g + fk;
                        f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                        sC[ml * CH + (n4 ^ (ml & 7))] = v;
                    }
            }
        }
        // LDS-only barriers in this loop: a __syncthreads() also waits for the previous pass's output stores (~2 us of HBM write
        // latency per pass, per tile of a few microseconds of MFMA work)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // phase 2
        if constexpr (FAST != 0) {
            constexpr int NPS = (WM + RP - 1) / RP, RG = FAST == 2 ? (NPS % 4 == 0 ? 4 : NPS % 2 == 0 ? 2 : 1) : NPS;
#pragma unroll
            for (int g0 = 0; g0 < NPS; g0 += RG) {
                [[maybe_unused]] f16x8 rv[RG];
                if constexpr (FAST == 2) {
#pragma unroll
                    for (int q = 0; q < RG; ++q) {      // unconditional requests of clamped rows (a row beyond M repeats the last one)
                        int m = m0 + pass * WM + (g0 + q) * RP + r0;
                        m = m < p.M ? m : p.M - 1;
                        rv[q] = *reinterpret_cast<const f16x8*>(p.res + (size_t)m * p.res_pix_stride + n);
                    }
                }
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const int ml = (g0 + q) * RP + r0;
                    const int m = m0 + pass * WM + ml;
                    if (ml >= WM) continue;
                    const f32x4 v0 = sC[ml * CH + ((2 * n8) ^ (ml & 7))];
                    const f32x4 v1 = sC[ml * CH + ((2 * n8 + 1) ^ (ml & 7))];
                    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    f16x8 hv;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float t = fmaf(v[k], sc[k], sh[k]);
                        t = p.leaky ? (t > 0.f ? t : t * 0.1f) : t;
                        if constexpr (FAST == 2) t += (float)rv[q][k];
                        hv[k] = (_Float16)t;
                    }
                    if (m < p.M) *reinterpret_cast<f16x8*>(static_cast<_Float16*>(p.out) + (size_t)m * p.out_pix_stride + n) = hv;
                }
            }
        } else if (p.out_mode != 2) {
#pragma unroll 2
            for (int ps = 0; ps < (WM + RP - 1) / RP; ++ps) {
                const int ml = ps * RP + r0;
                const int m = m0 + pass * WM + ml;
                if (ml >= WM || m >= p.M || nvalid <= 0) continue;
                const f32x4 v0 = sC[ml * CH + ((2 * n8) ^ (ml & 7))];
                const f32x4 v1 = sC[ml * CH + ((2 * n8 + 1) ^ (ml & 7))];
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = fmaf(v[k], sc[k], sh[k]);
                    v[k] = p.leaky ? (t > 0.f ? t : t * 0.1f) : t;
                }
                if (p.out_f32) {
                    float* o = static_cast<float*>(p.out) + (size_t)m * p.out_pix_stride + n;
                    if (vec) {
                        *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = v[k];
                    }
                    continue;
                }
                _Float16* const outh = static_cast<_Float16*>(p.out);
                if (p.out_mode == 0) {
                    _Float16* o = outh + (size_t)m * p.out_pix_stride + n;
                    if (p.res) {
                        const _Float16* rp = p.res + (size_t)m * p.res_pix_stride + n;
                        if (vec) {
                            const f16x8 rv = *reinterpret_cast<const f16x8*>(rp);
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
                        } else {
                            for (int k = 0; k < 8 && k < nvalid; ++k) v[k] += (float)rp[k];
                        }
                    }
                    if (vec) {
                        f16x8 hv;
#pragma unroll
                        for (int k = 0; k < 8; ++k) hv[k] = (_Float16)v[k];
                        *reinterpret_cast<f16x8*>(o) = hv;
                    } else {
                        for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = (_Float16)v[k];
                    }
                } else {
                    f16x8 hv;
#pragma unroll
                    for (int k = 0; k < 8; ++k) hv[k] = (_Float16)v[k];
                    const int bi = m / p.HoWo;
                    const int rr = m - bi * p.HoWo;
                    const int oy = rr / p.Wo, ox = rr - oy * p.Wo;
                    const int Wu = p.Wo * p.up;
                    const size_t base = ((size_t)bi * p.Ho * p.up + (size_t)oy * p.up) * Wu + (size_t)ox * p.up;
                    for (int dy = 0; dy < p.up; ++dy)
                        for (int dx = 0; dx < p.up; ++dx) {
                            _Float16* o = outh + (base + (size_t)dy * Wu + dx) * p.out_pix_stride + n;
                            if (vec) *reinterpret_cast<f16x8*>(o) = hv;
                            else
                                for (int k = 0; k < 8 && k < nvalid; ++k) o[k] = hv[k];
                        }
                }
            }
        } else {
            // NCHW fp32 output (orientation head): consecutive threads walk pixels of one channel
            const float* sCf = reinterpret_cast<const float*>(smem);
            float* const outf = static_cast<float*>(p.out);
            const int nch = min(BN, p.cout - n0);
            for (int idx = tid; idx < nch * WM; idx += NT) {
                const int nl = idx / WM, ml = idx - nl * WM;
                const int m = m0 + pass * WM + ml;
                if (m >= p.M) continue;
                const int nn = n0 + nl;
                float t = fmaf(sCf[(ml * CH + ((nl >> 2) ^ (ml & 7))) * 4 + (nl & 3)], p.scale[nn], p.shift[nn]);
                if (p.leaky) t = t > 0.f ? t : t * 0.1f;
                const int bi = m / p.HoWo;
                const int rr = m - bi * p.HoWo;
                outf[((size_t)bi * p.cout + nn) * p.HoWo + rr] = t;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the C rows are dead before the next pass / the next tile's
        __builtin_amdgcn_s_barrier();                            // operands land in LDS; the stores keep flying
    }
}

}  // namespace om
