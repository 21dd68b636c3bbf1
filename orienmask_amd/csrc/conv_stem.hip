// First layer of DarkNet-53 for gfx950: 3x3 stride-1 conv 3 -> 32, BN + LeakyReLU fused,
// reading the user's NCHW image and writing NHWC.
//
// Reference: backbone.conv1 = conv_bn_leaky(3, 32, 3, padding=1)
// (/root/reference/model/backbone/darknet.py:20, /root/reference/model/base.py:104-137).
//
// K = 27 is too short for the matrix cores and the layer is HBM-bound (reads 12 B, writes
// 128 B per pixel), so it runs on the vector ALUs:
//   * workgroup = a 32-pixel-wide column of STEM_STRIPS strips of 8 rows; per strip the 3 x 10 x 34 input patch
//     (zero padded) is staged in LDS, coalesced along x from the NCHW planes; the next strip's patch is fetched
//     into registers while the current one computes, and the 108 weights per thread are loaded once per column;
//   * thread = (pixel column, 4 output channels); its 27 x 4 weights live in registers;
//   * the 8 channel-quads of a pixel are 8 neighbouring lanes, so one wave store writes
//     8 pixels x 128 B = 1 KiB contiguous NHWC.
#include "om_common.h"

namespace om {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STEM_TX = 32, STEM_TY = 8, STEM_CO = 32;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((__vector_size__(4 * sizeof(unsigned))));
typedef unsigned u32x2s __attribute__((__vector_size__(2 * sizeof(unsigned))));

constexpr int STEM_STRIPS = 8;      // 8-row strips one workgroup walks down (weights and scale/shift loaded once)
constexpr int STEM_PATCH = 3 * (STEM_TY + 2) * (STEM_TX + 2);
constexpr int STEM_LD = (STEM_PATCH + 255) / 256;     // patch elements each thread stages

// OutT = float: the f32 path; OutT = _Float16: the fp16-activation path (same arithmetic, rounded once at the store)
template <typename OutT>
__global__ __launch_bounds__(256) void conv_stem_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, OutT* __restrict__ out,
                                                        int H, int W) {
    __shared__ float patch[3][STEM_TY + 2][STEM_TX + 2];
    float* const patch_flat = &patch[0][0][0];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * STEM_TX, b = blockIdx.z;
    const float* img = in + (size_t)b * 3 * H * W;
    // Every request goes through a buffer descriptor of THIS image's planes / output rows with an out-of-range offset where the
    // pixel does not exist (zeros for the padding, dropped stores), not through a branch: loads and stores retire through one
    // in-order counter, and with conditional requests the compiler's wait for the next strip's patch was vmcnt(0) -- every strip
    // waited for the round trip of the previous strip's eight stores (conv_wino14.hip's epilogue).
    const auto rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, 3 * H * W * 4, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * H * W * STEM_CO, 0, H * W * STEM_CO * (int)sizeof(OutT), 0x00020000);
    // the next strip's patch elements travel through registers while the current strip computes
    float stage[STEM_LD];
    auto fetch = [&](int y0) {
#pragma unroll
        for (int i = 0; i < STEM_LD; ++i) {
            const int e = tid + i * 256;
            const int c = e / ((STEM_TY + 2) * (STEM_TX + 2));
            const int r = e - c * (STEM_TY + 2) * (STEM_TX + 2);
            const int py = r / (STEM_TX + 2), px = r - py * (STEM_TX + 2);
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            const bool ok = e < STEM_PATCH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, ok ? ((c * H + gy) * W + gx) * 4 : (int)0x80000000, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < STEM_LD; ++i) {
            const int e = tid + i * 256;
            if (e < STEM_PATCH) patch_flat[e] = stage[i];
        }
    };
    const int strip0 = blockIdx.y * STEM_STRIPS;
    fetch(strip0 * STEM_TY);
    const int quad = tid & 7, px = tid >> 3;
    // weights [cout][tap=(kh*3+kw)][ci=3] -> this thread's 4 output channels
    f32x4 wr[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        wr[t][0] = w[(quad * 4 + 0) * 27 + t];
        wr[t][1] = w[(quad * 4 + 1) * 27 + t];
        wr[t][2] = w[(quad * 4 + 2) * 27 + t];
        wr[t][3] = w[(quad * 4 + 3) * 27 + t];
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + quad * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + quad * 4);
    // the weights and scale / shift have arrived BEFORE the loop as far as the compiler's wait bookkeeping is concerned: pending at
    // the loop's entry, they made the first use inside it `vmcnt(0)` in every iteration -- a wait for the patch requested a moment ago
#pragma unroll
    for (int t = 0; t < 27; ++t) asm volatile("" ::"v"(wr[t]));
    asm volatile("" ::"v"(sc), "v"(sh));
    for (int si = 0; si < STEM_STRIPS; ++si) {
        const int y0 = (strip0 + si) * STEM_TY;
        if (y0 >= H) break;
        commit();
        __syncthreads();
        if (si + 1 < STEM_STRIPS && y0 + STEM_TY < H) fetch(y0 + STEM_TY);
#pragma unroll      // all eight rows: the stores behind the next strip's requests are then counted exactly (vmcnt(8 + ...))
        for (int ty = 0; ty < STEM_TY; ++ty) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const float v = patch[ci][ty + kh][px + kw];
                        const f32x4 ww = wr[(kh * 3 + kw) * 3 + ci];
                        acc[0] = fmaf(v, ww[0], acc[0]);
                        acc[1] = fmaf(v, ww[1], acc[1]);
                        acc[2] = fmaf(v, ww[2], acc[2]);
                        acc[3] = fmaf(v, ww[3], acc[3]);
                    }
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = fmaf(acc[k], sc[k], sh[k]);
                o[k] = v > 0.f ? v : v * 0.1f;
            }
            const int gy = y0 + ty, gx = x0 + px;
            const int off = gy < H && gx < W ? ((gy * W + gx) * STEM_CO + quad * 4) * (int)sizeof(OutT) : (int)0x80000000;
            if constexpr (sizeof(OutT) == 4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, o), rs_out, off, 0, 0);
            } else {
                const f16x4 h = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, h), rs_out, off, 0, 0);
            }
        }
        __syncthreads();      // everyone is done with the patch before the next strip overwrites it
    }
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

int launch_zero_words(void* ptr, size_t n_words, hipStream_t stream) {
    OM_REQUIRE(ptr && (reinterpret_cast<uintptr_t>(ptr) & 3) == 0, OM_EINVAL, "zero_words: null or unaligned pointer");
    if (n_words == 0) return OM_OK;
    const size_t blocks = (n_words + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream,
                       static_cast<unsigned*>(ptr), n_words);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int launch_conv_stem(const float* in_nchw, int B, int H, int W, const float* w, const float* scale,
                     const float* shift, int cout, float* out_nhwc, hipStream_t stream) {
    OM_REQUIRE(in_nchw && w && scale && shift && out_nhwc, OM_EINVAL, "stem: null pointer");
    OM_REQUIRE(cout == STEM_CO, OM_EINVAL, "stem: cout=%d, only 32 supported", cout);
    OM_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0, OM_EINVAL, "stem: bad shape B=%d H=%d W=%d", B, H, W);
    OM_REQUIRE((long long)H * W * STEM_CO * 4 < 0x7FFFFFF0ll, OM_EINVAL, "stem: an image of %d x %d exceeds a buffer descriptor", H, W);
    dim3 grid((W + STEM_TX - 1) / STEM_TX, (H + STEM_TY * STEM_STRIPS - 1) / (STEM_TY * STEM_STRIPS), B);
    hipLaunchKernelGGL(conv_stem_kernel<float>, grid, dim3(256), 0, stream, in_nchw, w, scale, shift, out_nhwc, H, W);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int launch_conv_stem_f16(const float* in_nchw, int B, int H, int W, const float* w, const float* scale,
                         const float* shift, int cout, void* out_nhwc_f16, hipStream_t stream) {
    OM_REQUIRE(in_nchw && w && scale && shift && out_nhwc_f16, OM_EINVAL, "stem: null pointer");
    OM_REQUIRE(cout == STEM_CO, OM_EINVAL, "stem: cout=%d, only 32 supported", cout);
    OM_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0, OM_EINVAL, "stem: bad shape B=%d H=%d W=%d", B, H, W);
    OM_REQUIRE((long long)H * W * STEM_CO * 4 < 0x7FFFFFF0ll, OM_EINVAL, "stem: an image of %d x %d exceeds a buffer descriptor", H, W);
    dim3 grid((W + STEM_TX - 1) / STEM_TX, (H + STEM_TY * STEM_STRIPS - 1) / (STEM_TY * STEM_STRIPS), B);
    hipLaunchKernelGGL(conv_stem_kernel<_Float16>, grid, dim3(256), 0, stream, in_nchw, w, scale, shift,
                       static_cast<_Float16*>(out_nhwc_f16), H, W);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // namespace om
