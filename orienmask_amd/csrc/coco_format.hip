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
// counts (a few KB per mask) ever leave the device.
//
// Round 4: this stage cost 5 ms per image -- ten times forward + postprocess -- for two reasons.
//   * recover_rle_kernel walked the OUTPUT in column-major order, so every lane gathered single bytes of the row-major
//     source a whole row apart (four dependent byte gathers per pixel, 75 workgroup scans per mask).
//     recover_rle_lds_kernel evaluates the resize in ROW-major order (adjacent lanes read adjacent source bytes), packs each
//     thread's 32 consecutive rows of one output column into a word of a column-major BITMAP in LDS (38 KB for 480 x 640),
//     and finds the transitions with word arithmetic: one workgroup scan per mask.  The old kernel stays for images whose
//     bitmap does not fit LDS.
//   * pycocotools' LEB128-like string packing of the counts ran in a Python loop.  It is now the kernels' last phase
//     (rle_string_phase): each run's characters are sized, a workgroup scan gives their offsets, the workgroup reserves its
//     span of ONE byte buffer shared by every mask of the batch (an atomic cursor), and the host copies the used part once.
//
// Built with -ffp-contract=off: the float arithmetic follows torch's operation order (source index by fmaf,
// taps combined as in preprocess.hip), so the resized masks are bit-identical to the reference's.
#include <atomic>
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

// ---- round 4: strings on the device, and the resize evaluated where the source is coalesced -------------------------
struct RleStringOut {
    uint8_t* bytes;          // one buffer for every mask of the batch; NULL: no strings wanted
    long long capacity;
    int32_t* cursor;         // [0] bytes reserved so far (atomic), [1] masks whose string did not fit (atomic)
    int32_t* off;            // [K] offset of mask k's string in `bytes` (-1: did not fit)
    int32_t* len;            // [K] its length
};

// Workgroup-wide exclusive scan of one int per thread (RLE_THREADS threads); returns the exclusive prefix, *total the sum.
// s_wave: RLE_THREADS / 64 ints of LDS.  Two barriers; safe to call back to back.
__device__ __forceinline__ int block_scan_excl(int v, int* s_wave, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
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
    *total = tot;
    return off + x - v;
}

// pycocotools rleToString (common/maskApi.c; restated for the checker in oracle/rle_ref.c): from the fourth run on the value
// written is the difference to the run two back; 5 data bits per char, bit 0x20 = continuation, sign carried by bit 0x10,
// offset 48.  counts[0..m) are this mask's run lengths in global memory, complete and visible to the whole workgroup.
__device__ __forceinline__ int rle_run_chars(long long x, uint8_t* dst) {
    int n = 0;
    bool more = true;
    while (more) {
        int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        if (dst) dst[n] = (uint8_t)(c + 48);
        ++n;
    }
    return n;
}

__device__ void rle_string_phase(const uint32_t* counts, int m, const RleStringOut& so, int k, int* s_wave, int* s_bcast) {
    const int tid = threadIdx.x;
    const int rpt = (m + RLE_THREADS - 1) / RLE_THREADS;            // a thread's runs are consecutive: offsets follow run order
    const int i0 = tid * rpt, i1 = min(m, i0 + rpt);
    int nch = 0;
    for (int i = i0; i < i1; ++i) {
        long long x = (long long)counts[i];
        if (i > 2) x -= (long long)counts[i - 2];
        nch += rle_run_chars(x, nullptr);
    }
    int total = 0;
    const int off = block_scan_excl(nch, s_wave, &total);
    if (tid == 0) {
        long long base = (long long)atomicAdd(so.cursor, total);
        if (base + total > so.capacity || base < 0) { atomicAdd(so.cursor + 1, 1); base = -1; }
        so.off[k] = (int32_t)base;
        so.len[k] = total;
        s_bcast[0] = (int)base;
    }
    __syncthreads();
    const int base = s_bcast[0];
    if (base < 0) return;
    uint8_t* dst = so.bytes + base + off;
    for (int i = i0; i < i1; ++i) {
        long long x = (long long)counts[i];
        if (i > 2) x -= (long long)counts[i - 2];
        dst += rle_run_chars(x, dst);
    }
}

// positions of the value changes (already in counts[0..n)) -> run lengths, in place: c[0] = T[0], c[i] = T[i] - T[i-1],
// c[n] = total - T[n-1].  Chunks from the top down so that counts[i-1] of a chunk's first element is still a position.
__device__ __forceinline__ void rle_positions_to_lengths(uint32_t* counts, int n, long long total) {
    const int tid = threadIdx.x;
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

// One workgroup per mask.  Dynamic LDS: [oh] row taps (int2 + float2) | [ow * wpc] bitmap words, column-major: word
// x * wpc + yb holds rows 32 yb .. 32 yb + 31 of output column x (bit i = row 32 yb + i), i.e. pycocotools' pixel order.
// A batch of images in one launch (their masks are independent workgroups; image after image each launch would fill 100 of the
// chip's 256 CUs): the per-image parameters travel by value, blockIdx.x is the mask's index in the batch.
constexpr int RLE_BATCH = 24;
struct RleBatch {
    RleParams img[RLE_BATCH];      // counts / n_runs / resized already point at the image's first mask
    int first[RLE_BATCH + 1];      // index of the image's first mask; first[n] = number of masks
    int n;
};

__global__ __launch_bounds__(RLE_THREADS) void recover_rle_lds_kernel(const RleBatch bt, const RleStringOut so_all) {
    extern __shared__ __align__(16) unsigned char rle_smem[];
    __shared__ int s_wave[RLE_THREADS / 64];
    __shared__ int s_bcast[2];
    int im = 0;
    while (im + 1 < bt.n && (int)blockIdx.x >= bt.first[im + 1]) ++im;
    const RleParams& q = bt.img[im];
    const int k = blockIdx.x - bt.first[im], tid = threadIdx.x;
    RleStringOut so = so_all;
    if (so.bytes) { so.off += bt.first[im]; so.len += bt.first[im]; }
    const int wpc = (q.oh + 31) / 32;
    int4* s_tap = reinterpret_cast<int4*>(rle_smem);                 // y0, y1, bits of wy0, bits of wy1
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(rle_smem + (size_t)q.oh * sizeof(int4));
    const uint8_t* m = q.mask + (size_t)k * q.H * q.W;
    uint32_t* counts = q.counts + (size_t)k * q.max_runs;
    const long long total = (long long)q.oh * q.ow;
    const int nwords = q.ow * wpc;

    for (int y = tid; y < q.oh; y += RLE_THREADS) {
        int y0, y1;
        float wy0, wy1;
        tap(y, q.scale_h, q.ch, y0, y1, wy0, wy1);
        if (q.vflip) { y0 = q.ch - 1 - y0; y1 = q.ch - 1 - y1; }   // flips act on the cropped mask (coco_eval.py:198-201)
        s_tap[y] = make_int4((q.crop_top + y0) * q.W + q.crop_left, (q.crop_top + y1) * q.W + q.crop_left,
                             __float_as_int(wy0), __float_as_int(wy1));
    }
    __syncthreads();
    // ---- phase A: the resize, row-major (adjacent lanes = adjacent output columns = adjacent source bytes)
    for (int item = tid; item < nwords; item += RLE_THREADS) {
        const int yb = item / q.ow, x = item - yb * q.ow;
        int x0, x1;
        float wx0, wx1;
        tap(x, q.scale_w, q.cw, x0, x1, wx0, wx1);
        if (q.hflip) { x0 = q.cw - 1 - x0; x1 = q.cw - 1 - x1; }
        uint32_t word = 0;
        const int ylim = min(32, q.oh - 32 * yb);
        // eight rows per step, every byte load of the step in flight before the first is used (no store in this loop: a store
        // through q.resized could alias the mask as far as the compiler knows, and would serialise the loads)
        for (int i0 = 0; i0 < ylim; i0 += 8) {
            uint8_t a00[8], a01[8], a10[8], a11[8];
            int4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t[j] = s_tap[min(32 * yb + i0 + j, q.oh - 1)];
                const uint8_t* r0 = m + t[j].x;
                const uint8_t* r1 = m + t[j].y;
                a00[j] = r0[x0]; a01[j] = r0[x1]; a10[j] = r1[x0]; a11[j] = r1[x1];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float top = fmaf((float)a00[j], wx0, (float)a01[j] * wx1);
                const float bot = fmaf((float)a10[j], wx0, (float)a11[j] * wx1);
                const float v = fmaf(top, __int_as_float(t[j].z), bot * __int_as_float(t[j].w));
                const int bit = (int)rintf(v);                      // torch.round: half to even
                word |= (i0 + j < ylim ? (uint32_t)(bit & 1) : 0u) << ((i0 + j) & 31);
            }
        }
        if (q.resized)
            for (int i = 0; i < ylim; ++i) q.resized[((size_t)k * q.oh + 32 * yb + i) * q.ow + x] = (uint8_t)((word >> i) & 1u);
        s_bits[x * wpc + yb] = word;
    }
    __syncthreads();
    // ---- phase B: value changes.  A thread owns wpt consecutive words of the column-major bitmap.
    const int wpt = (nwords + RLE_THREADS - 1) / RLE_THREADS;
    const int w0 = tid * wpt, w1 = min(nwords, w0 + wpt);
    const int last_bits = q.oh - 32 * (wpc - 1);                     // valid bits of a column's last word (1..32)
    auto changes = [&](int wi) -> uint32_t {
        const int x = wi / wpc, yb = wi - x * wpc;
        const uint32_t w = s_bits[wi];
        uint32_t prev;
        if (yb > 0) prev = s_bits[wi - 1] >> 31;                     // a column's inner words are full
        else prev = x > 0 ? (s_bits[wi - 1] >> (last_bits - 1)) & 1u : 0u;
        const uint32_t valid = (yb == wpc - 1 && last_bits < 32) ? ((1u << last_bits) - 1u) : 0xFFFFFFFFu;
        return (w ^ ((w << 1) | prev)) & valid;
    };
    int nflag = 0;
    for (int wi = w0; wi < w1; ++wi) nflag += __popc(changes(wi));
    int n = 0;
    int slot = block_scan_excl(nflag, s_wave, &n);
    for (int wi = w0; wi < w1; ++wi) {
        uint32_t c = changes(wi);
        const int x = wi / wpc, yb = wi - x * wpc;
        const uint32_t pos0 = (uint32_t)((long long)x * q.oh + 32 * yb);
        while (c) {
            const int b = __ffs(c) - 1;
            c &= c - 1;
            if (slot < q.max_runs) counts[slot] = pos0 + b;           // positions first, lengths below
            ++slot;
        }
    }
    if (tid == 0) q.n_runs[k] = n + 1;
    if (n + 1 > q.max_runs) {                                         // the caller retries this mask with a larger buffer
        if (so.bytes && tid == 0) { so.off[k] = -1; so.len[k] = 0; }
        return;
    }
    __threadfence_block();
    __syncthreads();
    rle_positions_to_lengths(counts, n, total);
    if (so.bytes) {
        __threadfence_block();
        __syncthreads();
        rle_string_phase(counts, n + 1, so, k, s_wave, s_bcast);
    }
}

// the string phase alone, after recover_rle_kernel (images whose bitmap does not fit LDS)
__global__ __launch_bounds__(RLE_THREADS) void rle_string_kernel(const uint32_t* counts_all, const int32_t* n_runs, int max_runs,
                                                                 const RleStringOut so) {
    __shared__ int s_wave[RLE_THREADS / 64];
    __shared__ int s_bcast[2];
    const int k = blockIdx.x;
    const int m = n_runs[k];
    if (m > max_runs) {
        if (threadIdx.x == 0) { so.off[k] = -1; so.len[k] = 0; }
        return;
    }
    rle_string_phase(counts_all + (size_t)k * max_runs, m, so, k, s_wave, s_bcast);
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

static int fill_rle_params(om::RleParams& q, const uint8_t* mask, int K, int H, int W, int crop_top, int crop_down, int crop_left,
                           int crop_right, int hflip, int vflip, int orig_h, int orig_w, uint32_t* counts, int max_runs,
                           int32_t* n_runs, uint8_t* resized_or_null, const char* who) {
    OM_REQUIRE(mask && counts && n_runs && K > 0 && H > 0 && W > 0 && orig_h > 0 && orig_w > 0 && max_runs >= 1, OM_EINVAL,
               "%s: bad argument", who);
    OM_REQUIRE(crop_top >= 0 && crop_down >= 0 && crop_left >= 0 && crop_right >= 0 && crop_top + crop_down < H &&
                   crop_left + crop_right < W,
               OM_EINVAL, "%s: crop (%d,%d,%d,%d) leaves nothing of %dx%d", who, crop_top, crop_down, crop_left,
               crop_right, H, W);
    OM_REQUIRE((long long)orig_h * orig_w < (1ll << 31), OM_EINVAL, "%s: %d x %d pixels exceed the 32-bit run lengths", who, orig_h, orig_w);
    q.mask = mask; q.counts = counts; q.n_runs = n_runs; q.resized = resized_or_null;
    q.K = K; q.H = H; q.W = W;
    q.crop_top = crop_top; q.crop_left = crop_left; q.ch = H - crop_top - crop_down; q.cw = W - crop_left - crop_right;
    q.hflip = hflip; q.vflip = vflip; q.oh = orig_h; q.ow = orig_w; q.max_runs = max_runs;
    q.scale_h = (float)q.ch / (float)orig_h;
    q.scale_w = (float)q.cw / (float)orig_w;
    return OM_OK;
}

static size_t rle_lds_bytes(int orig_h, int orig_w) { return (size_t)orig_h * 16 + (size_t)orig_w * ((orig_h + 31) / 32) * 4; }
// Most LDS one mask's column-major bitmap may take (480 x 640: 46 KB with the row taps): what the device reports per workgroup
// (163 840 B on MI355X) less 10 KiB for the kernel's static arrays.  Queried once per process (every GPU of a node is the same
// part); -1: not yet known.
static size_t rle_lds_max() {
    static std::atomic<long long> cached{-1};
    long long v = cached.load(std::memory_order_acquire);
    if (v < 0) {
        int dev = 0, per_block = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess ||
            per_block < 64 * 1024)
            per_block = 64 * 1024;              // the architectural default: no attribute needed below it
        v = per_block - 10 * 1024;
        cached.store(v, std::memory_order_release);
    }
    return (size_t)v;
}

// images [i0, i1) of a batch whose bitmaps fit LDS: one launch
static int launch_rle_lds_batch(om::RleBatch& bt, size_t lds, const om::RleStringOut& so, hipStream_t st) {
    // more than the default 64 KiB of dynamic LDS needs the attribute on EVERY device that launches the kernel: set per launch
    // (a host-side table update, no device work; a process-wide "done" flag would cover the first GPU only and race between threads)
    if (lds > 64 * 1024)
        OM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(om::recover_rle_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)rle_lds_max()));
    hipLaunchKernelGGL(om::recover_rle_lds_kernel, dim3(bt.first[bt.n]), dim3(om::RLE_THREADS), lds, st, bt, so);
    OM_CHECK_HIP(hipGetLastError());
    return OM_OK;
}

static int launch_recover_rle(const uint8_t* mask, int K, int H, int W, int crop_top, int crop_down, int crop_left, int crop_right,
                              int hflip, int vflip, int orig_h, int orig_w, uint32_t* counts, int max_runs, int32_t* n_runs,
                              uint8_t* resized_or_null, const om::RleStringOut& so, om_stream stream, const char* who) {
    if (K == 0) return OM_OK;
    om::RleBatch bt;
    if (int rc = fill_rle_params(bt.img[0], mask, K, H, W, crop_top, crop_down, crop_left, crop_right, hflip, vflip, orig_h, orig_w,
                                 counts, max_runs, n_runs, resized_or_null, who))
        return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = rle_lds_bytes(orig_h, orig_w);
    if (lds <= rle_lds_max()) {
        bt.n = 1; bt.first[0] = 0; bt.first[1] = K;
        return launch_rle_lds_batch(bt, lds, so, st);
    }
    hipLaunchKernelGGL(om::recover_rle_kernel, dim3(K), dim3(om::RLE_THREADS), 0, st, bt.img[0]);
    OM_CHECK_HIP(hipGetLastError());
    if (so.bytes) {
        hipLaunchKernelGGL(om::rle_string_kernel, dim3(K), dim3(om::RLE_THREADS), 0, st, counts, n_runs, max_runs, so);
        OM_CHECK_HIP(hipGetLastError());
    }
    return OM_OK;
}

int om_recover_masks_rle(const uint8_t* mask, int K, int H, int W, int crop_top, int crop_down, int crop_left,
                         int crop_right, int hflip, int vflip, int orig_h, int orig_w, uint32_t* counts, int max_runs,
                         int32_t* n_runs, uint8_t* resized_or_null, om_stream stream) {
    om::RleStringOut so{nullptr, 0, nullptr, nullptr, nullptr};
    return launch_recover_rle(mask, K, H, W, crop_top, crop_down, crop_left, crop_right, hflip, vflip, orig_h, orig_w, counts,
                              max_runs, n_runs, resized_or_null, so, stream, "om_recover_masks_rle");
}

int om_recover_masks_rle_strings(const om_rle_image* images, int n_images, uint32_t* counts, int max_runs, int32_t* n_runs,
                                 uint8_t* str_bytes, long long str_capacity, int32_t* str_cursor, int32_t* str_off,
                                 int32_t* str_len, om_stream stream) {
    if (n_images == 0) return OM_OK;
    OM_REQUIRE(images && n_images > 0 && counts && n_runs && str_bytes && str_cursor && str_off && str_len && str_capacity > 0 &&
                   str_capacity < (1ll << 31) && max_runs >= 1,
               OM_EINVAL, "om_recover_masks_rle_strings: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const om::RleStringOut so{str_bytes, str_capacity, str_cursor, str_off, str_len};
    om::RleBatch bt;
    bt.n = 0;
    size_t lds = 0;
    int base = 0;          // masks before the images collected in bt
    int first = 0;         // masks before image i
    auto flush = [&]() -> int {
        if (bt.n == 0) return OM_OK;
        om::RleStringOut s2 = so;
        s2.off += base; s2.len += base;
        int rc = launch_rle_lds_batch(bt, lds, s2, st);
        base += bt.first[bt.n];
        bt.n = 0; lds = 0;
        return rc;
    };
    for (int i = 0; i < n_images; ++i) {
        const om_rle_image& im = images[i];
        OM_REQUIRE(im.K >= 0, OM_EINVAL, "om_recover_masks_rle_strings: image %d has K = %d", i, im.K);
        if (im.K == 0) continue;
        const size_t need = rle_lds_bytes(im.orig_h, im.orig_w);
        uint32_t* cnt = counts + (size_t)first * max_runs;
        if (need > rle_lds_max()) {            // a bitmap too large for LDS: this image alone, through the old kernel
            if (int rc = flush()) return rc;
            om::RleStringOut s2 = so;
            s2.off += first; s2.len += first;
            if (int rc = launch_recover_rle(im.mask, im.K, im.H, im.W, im.crop_top, im.crop_down, im.crop_left, im.crop_right, im.hflip,
                                            im.vflip, im.orig_h, im.orig_w, cnt, max_runs, n_runs + first, nullptr, s2, stream,
                                            "om_recover_masks_rle_strings"))
                return rc;
            base = first + im.K;
        } else {
            if (bt.n == om::RLE_BATCH) { if (int rc = flush()) return rc; }
            if (bt.n == 0) { base = first; bt.first[0] = 0; }
            if (int rc = fill_rle_params(bt.img[bt.n], im.mask, im.K, im.H, im.W, im.crop_top, im.crop_down, im.crop_left, im.crop_right,
                                         im.hflip, im.vflip, im.orig_h, im.orig_w, cnt, max_runs, n_runs + first, nullptr,
                                         "om_recover_masks_rle_strings"))
                return rc;
            bt.first[bt.n + 1] = bt.first[bt.n] + im.K;
            ++bt.n;
            lds = need > lds ? need : lds;
        }
        first += im.K;
    }
    return flush();
}

}  // extern "C"
