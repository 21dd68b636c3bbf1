// Output side of the path: COCOMetrics' conversion of one image's detections to
// the original image geometry, and the run-length encoding of its masks, on the GPU.
//
//   recover_bbox_kernel   COCOMetrics._recover_shape_bbox   /root/reference/eval/coco_eval.py:146-189
//   recover_rle_kernel    COCOMetrics._recover_shape_segm   /root/reference/eval/coco_eval.py:191-205
//                         (crop the paddings, flips, F.interpolate bilinear align_corners=False to the original size,
//                         round-half-even, uint8) + the column-major run lengths pycocotools' rleEncode produces
//                         for maskUtils.encode(np.asfortranarray(mask)) (coco_eval.py:120-122)
//
// The reference copies [K,544,544] booleans to the host (~30 MB per image), resizes on the CPU and calls
// pycocotools per mask; here one workgroup per mask evaluates the resize on the fly in column-major order,
// flags value changes, compacts their positions with a workgroup scan and emits the run lengths -- only the
// counts (a few KB per mask) ever leave the device.  pycocotools' LEB128-like string packing of the counts is
// done on the host (orienmask_amd/coco_format.py).
//
// Built with -ffp-contract=off: the float arithmetic follows torch's operation order (source index by fmaf,
// taps combined as in preprocess.hip), so the resized masks are bit-identical to the reference's.
#include "om_common.h"

namespace om {

struct RecoverBoxParams {
    const float* bbox;   // [K, stride] (cx, cy, w, h, ...), normalised
    float* out;          // [K,4] x, y, w, h in original-image pixels
    int K, stride;
    int has_cp, cp[6];   // collate_pad = (left, right, top, down, h, w)
    int has_p, p[6];     // pad         = (top, down, left, right, h, w)
    int hflip, vflip, oh, ow;
};

__global__ void recover_bbox_kernel(const RecoverBoxParams q) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= q.K) return;
    float bx = q.bbox[k * q.stride + 0], by = q.bbox[k * q.stride + 1];
    float bw = q.bbox[k * q.stride + 2], bh = q.bbox[k * q.stride + 3];
    if (q.has_cp) {
        const int left = q.cp[0], right = q.cp[1], top = q.cp[2], down = q.cp[3], h = q.cp[4], w = q.cp[5];
        const float nh = (float)(h - top - down), nw = (float)(w - left - right);
        bx = (bx * (float)w - (float)left) / nw;
        by = (by * (float)h - (float)top) / nh;
        bw = bw * (float)w / nw;
        bh = bh * (float)h / nh;
    }
    if (q.has_p) {
        const int top = q.p[0], down = q.p[1], left = q.p[2], right = q.p[3], h = q.p[4], w = q.p[5];
        const float nh = (float)(h - top - down), nw = (float)(w - left - right);
        bx = (bx * (float)w - (float)left) / nw;
        by = (by * (float)h - (float)top) / nh;
        bw = bw * (float)w / nw;
        bh = bh * (float)h / nh;
    }
    if (q.hflip) bx = 1.0f - bx;
    if (q.vflip) by = 1.0f - by;
    float* o = q.out + k * 4;
    o[0] = (bx - bw / 2.0f) * (float)q.ow;
    o[1] = (by - bh / 2.0f) * (float)q.oh;
    o[2] = bw * (float)q.ow;
    o[3] = bh * (float)q.oh;
}

struct RleParams {
    const uint8_t* mask;     // [K,H,W] 0/1
    uint32_t* counts;        // [K][max_runs]
    int32_t* n_runs;         // [K]   (> max_runs means the buffer was too small for that mask)
    uint8_t* resized;        // optional [K,oh,ow] row-major
    int K, H, W;
    int crop_top, crop_left, ch, cw;     // cropped region of the network-resolution mask
    int hflip, vflip, oh, ow, max_runs;
    float scale_h, scale_w;
};

__device__ __forceinline__ void tap(int d, float scale, int n_in, int& i0, int& i1, float& w0, float& w1) {
    float src = fmaf(scale, (float)d + 0.5f, -0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    w1 = src - (float)i0;
    w0 = 1.0f - w1;
}

constexpr int RLE_THREADS = 1024;

__global__ __launch_bounds__(RLE_THREADS) void recover_rle_kernel(const RleParams q) {
    __shared__ int s_wave[RLE_THREADS / 64];
    const int k = blockIdx.x, tid = threadIdx.x;
    const uint8_t* m = q.mask + (size_t)k * q.H * q.W;
    uint32_t* counts = q.counts + (size_t)k * q.max_runs;
    const long long total = (long long)q.oh * q.ow;

    auto value_at = [&](long long pos) -> int {       // pos = x * oh + y (column-major, pycocotools order)
        const int x = (int)(pos / q.oh), y = (int)(pos - (long long)x * q.oh);
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        tap(y, q.scale_h, q.ch, y0, y1, wy0, wy1);
        tap(x, q.scale_w, q.cw, x0, x1, wx0, wx1);
        // flips act on the cropped mask (coco_eval.py:198-201)
        if (q.vflip) { y0 = q.ch - 1 - y0; y1 = q.ch - 1 - y1; }
        if (q.hflip) { x0 = q.cw - 1 - x0; x1 = q.cw - 1 - x1; }
        const uint8_t* r0 = m + (size_t)(q.crop_top + y0) * q.W + q.crop_left;
        const uint8_t* r1 = m + (size_t)(q.crop_top + y1) * q.W + q.crop_left;
        const float top = fmaf((float)r0[x0], wx0, (float)r0[x1] * wx1);
        const float bot = fmaf((float)r1[x0], wx0, (float)r1[x1] * wx1);
        const float v = fmaf(top, wy0, bot * wy1);
        return (int)rintf(v);                          // torch.round: half to even
    };

    int run_base = 0;        // transitions found so far
    for (long long base = 0; base < total; base += RLE_THREADS * 4) {
        const long long p0 = base + (long long)tid * 4;
        int vals[5];
        vals[0] = (p0 == 0 || p0 > total) ? 0 : (p0 <= total ? value_at(p0 - 1) : 0);
        int flags = 0, nflag = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long pos = p0 + e;
            int v = 0;
            if (pos < total) {
                v = value_at(pos);
                if (q.resized) {
                    const int x = (int)(pos / q.oh), y = (int)(pos - (long long)x * q.oh);
                    q.resized[((size_t)k * q.oh + y) * q.ow + x] = (uint8_t)v;
                }
                if (v != vals[e]) { flags |= 1 << e; ++nflag; }
            }
            vals[e + 1] = v;
        }
        // workgroup exclusive scan of nflag
        const int lane = tid & 63, wave = tid >> 6;
        int x = nflag;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int yv = __shfl_up(x, d);
            if (lane >= d) x += yv;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < RLE_THREADS / 64; ++w) {
            const int t = s_wave[w];
            if (w < wave) off += t;
            tot += t;
        }
        __syncthreads();
        int slot = run_base + off + x - nflag;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (flags & (1 << e)) {
                if (slot < q.max_runs) counts[slot] = (uint32_t)(p0 + e);     // positions first, lengths below
                ++slot;
            }
        run_base += tot;
    }
    __syncthreads();
    // positions -> run lengths: c[0] = T[0], c[i] = T[i] - T[i-1], c[n] = total - T[n-1]
    const int n = run_base;
    if (tid == 0) q.n_runs[k] = n + 1;
    if (n + 1 > q.max_runs) return;
    // in-place difference needs the neighbour's ORIGINAL value: walk the chunks from the top down so that
    // counts[i-1] of a chunk's first element has not been rewritten yet, and split read / write by a barrier
    for (int b0 = (n / RLE_THREADS) * RLE_THREADS; b0 >= 0; b0 -= RLE_THREADS) {
        const int i = b0 + tid;
        uint32_t cur = 0, prev = 0;
        if (i <= n) {
            cur = i < n ? counts[i] : (uint32_t)total;
            prev = i > 0 ? counts[i - 1] : 0u;
        }
        __syncthreads();
        if (i <= n) counts[i] = cur - prev;
        __syncthreads();
    }
}

}  // namespace om

extern "C" {

int om_recover_bbox(const float* bbox, int K, int stride, const int32_t* collate_pad6, const int32_t* pad6, int hflip,
                    int vflip, int orig_h, int orig_w, float* out_xywh, om_stream stream) {
    if (K == 0) return OM_OK;
    OM_REQUIRE(bbox && out_xywh && K > 0 && stride >= 4 && orig_h > 0 && orig_w > 0, OM_EINVAL, "om_recover_bbox: bad argument");
    om::RecoverBoxParams q;
    q.bbox = bbox; q.out = out_xywh; q.K = K; q.stride = stride;
    q.has_cp = collate_pad6 != nullptr; q.has_p = pad6 != nullptr;
    for (int i = 0; i < 6; ++i) { q.cp[i] = collate_pad6 ? collate_pad6[i] : 0; q.p[i] = pad6 ? pad6[i] : 0; }
    q.hflip = hflip; q.vflip = vflip; q.oh = orig_h; q.ow = orig_w;
    hipLaunchKernelGGL(om::recover_bbox_kernel, dim3((K + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), q);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

int om_recover_masks_rle(const uint8_t* mask, int K, int H, int W, int crop_top, int crop_down, int crop_left,
                         int crop_right, int hflip, int vflip, int orig_h, int orig_w, uint32_t* counts, int max_runs,
                         int32_t* n_runs, uint8_t* resized_or_null, om_stream stream) {
    if (K == 0) return OM_OK;
    OM_REQUIRE(mask && counts && n_runs && K > 0 && H > 0 && W > 0 && orig_h > 0 && orig_w > 0 && max_runs >= 1, OM_EINVAL,
               "om_recover_masks_rle: bad argument");
    OM_REQUIRE(crop_top >= 0 && crop_down >= 0 && crop_left >= 0 && crop_right >= 0 && crop_top + crop_down < H &&
                   crop_left + crop_right < W,
               OM_EINVAL, "om_recover_masks_rle: crop (%d,%d,%d,%d) leaves nothing of %dx%d", crop_top, crop_down, crop_left,
               crop_right, H, W);
    om::RleParams q;
    q.mask = mask; q.counts = counts; q.n_runs = n_runs; q.resized = resized_or_null;
    q.K = K; q.H = H; q.W = W;
    q.crop_top = crop_top; q.crop_left = crop_left; q.ch = H - crop_top - crop_down; q.cw = W - crop_left - crop_right;
    q.hflip = hflip; q.vflip = vflip; q.oh = orig_h; q.ow = orig_w; q.max_runs = max_runs;
    q.scale_h = (float)q.ch / (float)orig_h;
    q.scale_w = (float)q.cw / (float)orig_w;
    hipLaunchKernelGGL(om::recover_rle_kernel, dim3(K), dim3(om::RLE_THREADS), 0, static_cast<hipStream_t>(stream), q);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

}  // extern "C"
