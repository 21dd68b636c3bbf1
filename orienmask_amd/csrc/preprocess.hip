// Input side of the path: the reference's GPU preprocessing
//   FastCOCOTransform.__call__: [n,h,w,c] -> permute(0,3,1,2) -> Resize (F.interpolate bilinear,
//   align_corners=False) -> Normalize ((x - mean) / std)          /root/reference/data/transform.py:444-510
//   pad(image, 32, 0): centred zero padding to a multiple of 32   /root/reference/infer.py:21-32
// fused into ONE kernel that reads the HWC image once and writes the padded NCHW network input once
// (the reference runs permute+contiguous, interpolate, sub_, div_, F.pad: five passes over the image).
//
// Bit-exactness: the source index is fmaf(scale, dst + 0.5, -0.5) and the taps are combined as
// fmaf(row(y0), wy0, row(y1) * wy1) with row(y) = fmaf(v[x0], wx0, v[x1] * wx1) -- the placement torch's
// CPU kernel compiles to (found by search, see oracle/orienmask_ref.py); this file is built with
// -ffp-contract=off so nothing else fuses.  Division by std is IEEE.
#include "om_common.h"

namespace om {

struct PreParams {
    const float* in;     // [N,h,w,3]
    float* out;          // [N,3,out_h,out_w]
    int N, h, w, rh, rw, out_h, out_w, pad_top, pad_left;
    float mean[3], stdv[3], pad_value, scale_h, scale_w;
};

__device__ __forceinline__ void bilinear_tap(int d, float scale, int n_in, int& i0, int& i1, float& w0, float& w1) {
    float src = fmaf(scale, (float)d + 0.5f, -0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    w1 = src - (float)i0;
    w0 = 1.0f - w1;
}

__global__ __launch_bounds__(256) void preprocess_kernel(const PreParams p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.N * p.out_h * p.out_w;
    if (idx >= total) return;
    const int x = (int)(idx % p.out_w);
    const long long r = idx / p.out_w;
    const int y = (int)(r % p.out_h);
    const int n = (int)(r / p.out_h);
    const size_t plane = (size_t)p.out_h * p.out_w;
    float* o = p.out + (size_t)n * 3 * plane + (size_t)y * p.out_w + x;
    const int ry = y - p.pad_top, rx = x - p.pad_left;
    if ((unsigned)ry >= (unsigned)p.rh || (unsigned)rx >= (unsigned)p.rw) {
        o[0] = p.pad_value; o[plane] = p.pad_value; o[2 * plane] = p.pad_value;
        return;
    }
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilinear_tap(ry, p.scale_h, p.h, y0, y1, wy0, wy1);
    bilinear_tap(rx, p.scale_w, p.w, x0, x1, wx0, wx1);
    const float* img = p.in + (size_t)n * p.h * p.w * 3;
    const float* p00 = img + ((size_t)y0 * p.w + x0) * 3;
    const float* p01 = img + ((size_t)y0 * p.w + x1) * 3;
    const float* p10 = img + ((size_t)y1 * p.w + x0) * 3;
    const float* p11 = img + ((size_t)y1 * p.w + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = fmaf(p00[c], wx0, p01[c] * wx1);
        const float bot = fmaf(p10[c], wx0, p11[c] * wx1);
        const float v = fmaf(top, wy0, bot * wy1);
        o[c * plane] = (v - p.mean[c]) / p.stdv[c];
    }
}

__global__ __launch_bounds__(256) void pad_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes,
                                                       int h, int w, int out_h, int out_w, int pad_top, int pad_left,
                                                       float pad_value) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * out_h * out_w) return;
    const int x = (int)(idx % out_w);
    const long long r = idx / out_w;
    const int y = (int)(r % out_h);
    const long long pl = r / out_h;
    const int sy = y - pad_top, sx = x - pad_left;
    float v = pad_value;
    if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) v = in[(pl * h + sy) * w + sx];
    out[idx] = v;
}

}  // namespace om

extern "C" {

int om_preprocess(const float* in_nhwc, int N, int h, int w, int resize_h, int resize_w, const float* mean3,
                  const float* std3, int pad_top, int pad_left, int out_h, int out_w, float pad_value, float* out_nchw,
                  om_stream stream) {
    OM_REQUIRE(in_nhwc && out_nchw && mean3 && std3, OM_EINVAL, "om_preprocess: null argument");
    OM_REQUIRE(N > 0 && h > 0 && w > 0 && resize_h > 0 && resize_w > 0, OM_EINVAL, "om_preprocess: bad shape");
    OM_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top + resize_h <= out_h && pad_left + resize_w <= out_w, OM_EINVAL,
               "om_preprocess: the resized image (%dx%d at +%d,+%d) does not fit the output %dx%d", resize_h, resize_w,
               pad_top, pad_left, out_h, out_w);
    om::PreParams p;
    p.in = in_nhwc; p.out = out_nchw;
    p.N = N; p.h = h; p.w = w; p.rh = resize_h; p.rw = resize_w; p.out_h = out_h; p.out_w = out_w;
    p.pad_top = pad_top; p.pad_left = pad_left; p.pad_value = pad_value;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
    p.scale_h = (float)h / (float)resize_h;      // area_pixel_compute_scale with size= given
    p.scale_w = (float)w / (float)resize_w;
    const long long total = (long long)N * out_h * out_w;
    hipLaunchKernelGGL(om::preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int om_pad_nchw(const float* in, long long planes, int h, int w, int pad_top, int pad_left, int out_h, int out_w,
                float pad_value, float* out, om_stream stream) {
    OM_REQUIRE(in && out && planes > 0 && h > 0 && w > 0, OM_EINVAL, "om_pad_nchw: bad argument");
    OM_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top + h <= out_h && pad_left + w <= out_w, OM_EINVAL,
               "om_pad_nchw: the image does not fit the output");
    const long long total = planes * out_h * out_w;
    hipLaunchKernelGGL(om::pad_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, out, planes, h, w, out_h, out_w, pad_top, pad_left, pad_value);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // extern "C"
