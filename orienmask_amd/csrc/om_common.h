// Internal helpers shared by the host-side translation units of liborienmask_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/orienmask_hip.h"

namespace om {

void set_error(const char* fmt, ...);

#define OM_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            om::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OM_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define OM_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            om::set_error(__VA_ARGS__);                                                 \
            return (code);                                                              \
        }                                                                               \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One fused convolution launch.  Pointers already include any channel offset of a view
// into a wider (concatenated) buffer; *_pix_stride is the float distance between pixels.
struct ConvArgs {
    const float* in;        // NHWC view [B,H,W,cin]
    const float* w;         // [cout_pad][ks*ks][cin]
    const float* scale;     // [cout_pad]
    const float* shift;     // [cout_pad]
    const float* res;       // optional NHWC view [B,Ho,Wo,cout], added after the activation
    float* out;
    int* ticket = nullptr;  // SYNC_WORDS device ints zeroed before the launch: [0] the tile / slot queue, [SK_FLAG_OFF + s] the
                            // "slot s has published" flags of the stream-K Winograd GEMM; nullptr = static grid
    float* sk_partial = nullptr;      // SK_PARTIAL_BYTES of device memory for stream-K partial tiles (nullptr: whole tiles only)
    hipEvent_t mid_event = nullptr;   // profiling: recorded between the two kernels of a Winograd layer
    int B, H, W, cin, in_pix_stride;
    int Ho, Wo, cout, cout_pad;
    int ks, stride;
    int leaky;
    int res_pix_stride;
    int out_pix_stride;     // NHWC modes
    int out_mode;           // 0: NHWC   1: NHWC, each output replicated up x up (nearest upsample)
                            // 2: NCHW contiguous [B,cout,Ho,Wo]
    int up;
    int split = 0;          // Winograd F(2x4) only: operands as hi/lo fp16 pairs (conv_wino24.hip)
    int* status = nullptr;  // the forward's status word (include/orienmask_hip.h: OM_STATUS_*), OR-ed by the split-operand kernels
                            // and the stream-K form; nullptr = not reported
    int ksplit_max = 0;     // conv_igemm_split.hip, launches of few tiles: most parts a tile's k loop may be cut into (needs sk_partial and
                            // the ticket's flag words; 0 / 1: whole tiles).  The sum order changes with the number of parts.
    int force_bm = 0, force_bn = 0;   // unit-test entries: tile shape of conv_igemm_split.hip for THIS call (0 = the chooser's)
    // gathered input (conv_igemm_split.hip, 1x1 layers): the cin channels are the concatenation of nseg tensors, segment g stored
    // at 1 / seg_up[g] of this layer's resolution and read nearest-up-sampled; nseg = 0: the plain view `in`
    int nseg = 0;
    const float* seg_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
    int seg_channels[4] = {0, 0, 0, 0}, seg_pix_stride[4] = {0, 0, 0, 0}, seg_up[4] = {1, 1, 1, 1};
};

constexpr int SK_SLOTS = 1024;                          // most resident workgroups of a stream-K launch (4 per CU)
constexpr int SK_FLAG_OFF = 16;
constexpr int SYNC_WORDS = SK_FLAG_OFF + SK_SLOTS;      // ints per launch
constexpr int STATUS_WORDS = 16;                        // behind the last layer's sync words: [0] the forward's status word
constexpr size_t SK_PARTIAL_BYTES = (size_t)512 * 32 * 256 * 16;   // Winograd F(2x4): 512 slots x 8 accumulators x 16 floats x 256 threads (64 MiB)

int launch_conv_igemm(const ConvArgs& a, hipStream_t stream);
// split-operand form (conv_igemm_split.hip): a.w = packed hi/lo fp16 weights, a.scale = scale * 2^-e
int launch_conv_igemm_split(const ConvArgs& a, hipStream_t stream);
void conv_tile_for_split(int M, int cout_pad, int* bm, int* bn);

// fp16-activation path (conv_igemm_f16.hip): in / w / res are fp16, scale / shift fp32, out fp16 unless out_f32.
struct ConvArgsH {
    const void* in;         // fp16 NHWC view [B,H,W,cin] (pixel stride in halfs)
    const void* w;          // fp16 [cout_pad][ks*ks][cin]
    const float* scale;
    const float* shift;
    const void* res;        // optional fp16 NHWC view
    void* out;
    int* ticket = nullptr;
    int B, H, W, cin, in_pix_stride;
    int Ho, Wo, cout, cout_pad;
    int ks, stride;
    int leaky;
    int res_pix_stride;
    int out_pix_stride;     // in elements of the output type
    int out_mode;           // 0: NHWC   1: NHWC replicated up x up (fp16 only)   2: NCHW contiguous (fp32 only)
    int up;
    int out_f32;            // 1: the output tensor is fp32 (the four head convolutions)
    // gathered input (conv_igemm_f16.hip, 1x1 layers; as ConvArgs'): nseg = 0: the plain view `in`
    int nseg = 0;
    const void* seg_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
    int seg_channels[4] = {0, 0, 0, 0}, seg_pix_stride[4] = {0, 0, 0, 0}, seg_up[4] = {1, 1, 1, 1};
};
int launch_conv_igemm_f16(const ConvArgsH& a, hipStream_t stream);     // dispatches stride-1 3x3 layers to conv3x3_f16.hip
int launch_conv3x3_f16(const ConvArgsH& a, hipStream_t stream);
bool conv3x3_f16_supported(const ConvArgsH& a);
void conv3x3_tile_for_f16(int M, int cout_pad, int W, int kc, int* bm, int* bn);
int conv3x3_f16_get_tall();
void conv3x3_f16_set_tall(int mode);      // conv3x3_f16.hip: 0 never / 1 where the tile chooser picks it (default) / 2 wherever it can run
void conv_tile_for_f16(int M, int cout_pad, int cin, int* bm, int* bn);
inline size_t conv_f16_weight_halfs(int cout_pad, int ks, int cin) {
    return (size_t)cout_pad * ks * ks * cin;
}
int launch_conv_stem_f16(const float* in_nchw, int B, int H, int W, const float* w, const float* scale,
                         const float* shift, int cout, void* out_nhwc_f16, hipStream_t stream);
void conv_tile_for(int M, int cout_pad, int* bm, int* bn);
// Winograd F(2x2,3x3) path (conv_wino.hip): a.w = U [16][cout_pad][cin]
int launch_conv_winograd(const ConvArgs& a, float* scratch, hipStream_t stream);
size_t wino_scratch_floats(int B, int H, int W, int C);
// Winograd F(2x4,3x3) path (conv_wino24.hip): a.w = U [24][cout_pad][cin]
int launch_conv_winograd24(const ConvArgs& a, float* scratch, hipStream_t stream);
size_t wino24_scratch_floats(int B, int H, int W, int C);
// fused F(4,3)-along-the-rows form with split operands (conv_wino14.hip): a.w = packed hi/lo weights [n_tiles][cin/16][6][3][64][32]
int launch_conv_wino14_split(const ConvArgs& a, hipStream_t stream);
// the same layer as two kernels with a 128 x 128 tile (conv_wino14.hip, round 6): V = the transformed input, written by a pre-pass into
// `scratch` (wino14_wide_scratch_floats floats), read by LDS-DMA; bit-identical to the fused kernel
size_t wino14_wide_scratch_floats(int B, int H, int W, int cin);
bool wino14_wide_supported(const ConvArgs& a);
bool wino14_wide_pays(const ConvArgs& a);
int launch_conv_wino14_wide(const ConvArgs& a, float* scratch, hipStream_t stream);
// backbone.conv1 + backbone.conv2.0 as one kernel with split operands (conv_stem2.hip): image NCHW -> conv2.0's NHWC output
// conv1 + conv2.0 of the fp16-activation configuration as one kernel (conv_stem2.hip: conv_stem2_f16_kernel)
int launch_conv_stem2_f16(const float* in_nchw, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                          const void* w2_f16, const float* scale2, const float* shift2, int cout2, int leaky2, void* out_nhwc_f16,
                          int out_pix_stride, hipStream_t stream);
// the optional third layer of conv_stem2_split_kernel: a 64 -> 32 1x1 convolution on conv2.0's outputs (conv_stem2.hip)
struct Stem2Third {
    const void* w_split;        // conv_weights_split rows [32][4][4][8] halfs
    const float* scale_split;   // scale * 2^-e
    const float* shift;
    float* out;                 // NHWC [B, H/2, W/2, out_pix_stride]
    int cout, leaky, out_pix_stride;
};
int launch_conv_stem2_split(const float* in_nchw, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                            const void* w2_split, const float* scale2_split, const float* shift2, int cout2, int leaky2,
                            float* out_nhwc, int out_pix_stride, int* status, hipStream_t stream, const Stem2Third* third = nullptr);
size_t wino14_weight_halfs(int cout_pad, int cin);
void wino14_geometry(int B, int H, int W, int* R, int* Ct, int* ncb, int* nrb);
int wino14_variant();
void wino14_set_variant(int v);      // conv_wino14.hip: 0 = the twelve-wave kernel (default), 1 = the dual-role kernel (conv_wino14d.hip) where it applies
bool wino_enabled();
int wino_bn(long long T, int cout_pad);   // N tile of the (unfused) Winograd GEMM at this size
bool wino_fused_for(int cin);     // true: the input transform is fused into the GEMM's loader   // tile shape launch_conv_igemm picks
// Clears n 32-bit words with a kernel.  The library never uses hipMemsetAsync: a memset node captured
// into a hipGraph was observed (ROCm 7.2, DESIGN.md "hipGraph") to leave part of the range uncleared on replay.
int launch_zero_words(void* ptr, size_t n_words, hipStream_t stream);
int launch_conv_stem(const float* in_nchw, int B, int H, int W, const float* w, const float* scale,
                     const float* shift, int cout, float* out_nhwc, hipStream_t stream);

}  // namespace om
