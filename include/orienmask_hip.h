/* orienmask_hip.h -- C ABI of the MI355X (gfx950) OrienMask inference hot path.
 *
 * One shared library, liborienmask_hip.so, plain pointers and sizes only (no torch types).
 * Every device buffer is allocated by the caller (the Python host passes
 * torch-ROCm tensor.data_ptr()); the library never allocates device memory on the hot path
 * and launches only on the stream it is handed.  All functions return 0 on success or a
 * negative OM_E* code; om_last_error() holds the message (thread-local).  No exceptions
 * cross this boundary; the library has no mutable global state besides that thread-local message, the process-wide
 * kernel-choice switches (om_set_*_variant, om_set_stem_fusion: A/B runs and tests) and the unit-test entries at the end, which
 * own a hipMalloc'ed tile-queue word each.
 * Threading: forwards of one om_model may be enqueued from several host threads when every thread uses its own stream and its
 * own workspace.  The caller serialises (a) stream captures of one model (they share one second stream), (b) om_profile_* and
 * (c) more than 64 caller streams with an attached postprocess at once.  The reference's plugin surface is single-threaded
 * under the GIL and never leaves this contract.
 *
 * What each entry point replaces in the reference (/root/reference):
 *   om_model_* / om_forward        OrienMaskYOLOFPNPlus.__init__/forward
 *                                  model/orienmask_yolo_fpnplus.py:9-90 (+ model/base.py:104-137,
 *                                  model/backbone/darknet.py:6-54), i.e. what trainer/tester.py:39-40
 *                                  and infer.py:154-155 call as `model(image)`
 *   om_postprocess                 OrienMaskYOLOPostProcess.apply
 *                                  eval/orienmask_yolo_postprocess.py:66-166, called at
 *                                  trainer/tester.py:43-44 and infer.py:156
 *   om_nms                         the pybind export `nms(dets[n,5], threshold) -> keep`
 *                                  eval/src/nms_cpu.cpp:65-75 / eval/src/nms_cuda.cpp:8-17, called
 *                                  from eval/function.py:69-72,98-101
 *   om_conv2d                      one ConvBNRelu (model/base.py:104-137); exported so parity tests
 *                                  can check a single layer against torch
 */
#ifndef ORIENMASK_HIP_H
#define ORIENMASK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OM_VERSION 140          /* 0.1.4: forward status word (om_forward_status_offset, OM_STATUS_*), no process-wide switches */

#define OM_OK 0
#define OM_EINVAL (-1)          /* bad argument (null pointer, shape not supported) */
#define OM_ESTATE (-2)          /* call order (forward before load_weights, ...) */
#define OM_ENOMEM (-3)          /* caller-provided workspace too small */
#define OM_EHIP (-4)            /* a HIP runtime call failed; see om_last_error() */

#define OM_MAX_SCALES 3
#define OM_MAX_ANCHORS 9

typedef struct om_model om_model;
typedef void* om_stream;        /* a hipStream_t (torch.cuda.current_stream().cuda_stream) */

/* Layout of one convolution inside the packed weight blob (all offsets in floats).
 * weights: [cout_pad][ksize*ksize][cin] (OHWI, rows >= cout zero);  scale/shift: [cout_pad]
 * y = conv(x) * scale + shift, then LeakyReLU(0.1) if leaky (BatchNorm folded by the host). */
typedef struct om_layer_info {
    char name[64];              /* reference module prefix, e.g. "backbone.conv4.3.conv.1" */
    int32_t cin, cout, cout_pad, ksize, stride;
    int32_t has_bn;             /* 1: conv(bias=False)+BN+leaky;  0: conv(bias=True) only */
    int32_t leaky;
    int32_t wino_planes;        /* 0: no Winograd weights; 16: F(2x2,3x3); 24: F(2x4,3x3) (2 rows x 4 columns per tile) */
    int64_t w_off, scale_off, shift_off;
    int64_t wino_off;           /* >= 0: Winograd weights U = G_y g G_x^T, [wino_planes][cout_pad][cin], plane index
                                   = 4 i + j (16) or 6 i + j (24), for the stride-1 3x3 layers; -1: none */
    int64_t wino_alt_off;       /* wino_planes == 24 only: the layer's F(2x2,3x3) weights [16][cout_pad][cin] as well, used when
                                   the batch is too small for the 2 x 4 tiling to fill the chip; -1 otherwise */
    int64_t w16_off;            /* fp16 path: offset IN HALFS into the fp16 weight blob (om_model_load_weights_f16):
                                   [cout_pad][ksize*ksize][cin] fp16 (rows >= cout zero); -1: the stem (always fp32) */
    int64_t wsplit_off;         /* split-operand mode (om_model_set_precision(m, 1)): offset in 4-BYTE WORDS into the split blob
                                   (om_model_load_weights_split) of the layer's weights as hi/lo fp16 pairs of
                                   x = weight * 2^e[cout]  (hi = fp16(x), lo = fp16(x - hi)):
                                   wino_planes == 24: the F(2x4,3x3) planes, [24][cout_pad][cin / 16][2][16] halfs -- per row and
                                     group of 16 input channels the 16 hi halfs, then the 16 lo halfs;
                                   otherwise: the direct weights, [cout_pad][ksize*ksize][cin / 16][4][8] halfs -- per group of
                                     16 input channels hi of channels {0-3, 8-11}, hi of {4-7, 12-15}, lo of {0-3, 8-11}, lo of
                                     {4-7, 12-15} (the order in which a lane half reads fp32 activations from LDS);
                                   -1: the stem (always fp32 operands) */
    int64_t wsplit_scale_off;   /* offset in 4-byte words into the split blob of [cout_pad] floats scale * 2^-e[cout] (the
                                   power of two is exact, so the epilogue rounds as with the unscaled weights) */
    int64_t wsplit_direct_off;  /* wino_planes == 24 only (else -1): the same layer's DIRECT 3x3 weights in the "otherwise" layout
                                   of wsplit_off, with exponents of their own, for the latency mode (om_model_set_latency_cells) */
    int64_t wsplit_direct_scale_off;   /* ... and their [cout_pad] floats scale * 2^-e[cout] */
} om_layer_info;

/* Constants of OrienMaskYOLOPostProcess.__init__ (eval/orienmask_yolo_postprocess.py:9-37). */
typedef struct om_post_cfg {
    int32_t num_scales;                         /* 3 */
    int32_t grid_h[OM_MAX_SCALES], grid_w[OM_MAX_SCALES];   /* coarse -> fine, e.g. 17,34,68 */
    int32_t image_h, image_w;                   /* 544, 544; orientation maps are image/4 */
    int32_t anchors_per_scale;                  /* 3 */
    float anchor_w[OM_MAX_ANCHORS], anchor_h[OM_MAX_ANCHORS];    /* pixels, index = anchor id */
    int32_t anchor_mask[OM_MAX_SCALES][3];      /* anchor ids used by each scale */
    int32_t num_classes;                        /* 80 */
    float conf_thresh;                          /* 0.005 */
    float nms_thresh;                           /* 0.5, suppress when IoU >= thresh */
    int32_t nms_pre, nms_post;                  /* 400, 100 */
    float orien_thresh;                         /* 0.3 */
    int32_t bbox_pix_stride;                    /* floats between pixels of the NHWC bbox heads
                                                   (256 for om_forward's outputs) */
    int32_t nms_semantics;                      /* which of the reference's two NMS backends (eval/function.py:98-101) the
                                                   suppression follows:
                                                   0 = nms_cpu (eval/src/nms_cpu.cpp:4-63): areas (x2-x1)*(y2-y1), suppress
                                                       when IoU >= nms_thresh, survivors in ascending candidate order;
                                                   1 = nms_cuda (eval/src/nms_kernel.cu:13-140): areas w*h, suppress when
                                                       IoU > nms_thresh, survivors in score-descending order */
    int32_t nms_normalized;                     /* batched_nms(normalized=...), eval/function.py:91-92: 1 = class offset
                                                   cls * 2.0; 0 = cls * (max(x, y) + max(w, h) / 2 + 0.5) over the image's
                                                   candidates */
    int32_t anchors_of_scale[OM_MAX_SCALES];    /* anchors of each scale where they differ (len(anchor_mask[i]),
                                                   postprocess.py:17-27), 1..3; 0 = anchors_per_scale.  num_scales may be 1..3:
                                                   the heads of unused scales are not read (pass NULL) */
} om_post_cfg;

int om_version(void);
const char* om_last_error(void);

/* ---- model ------------------------------------------------------------------------------- */
int om_model_create(om_model** out, int num_anchors, int num_classes);    /* OrienMaskYOLOFPNPlus */
/* variant 0: OrienMaskYOLOFPNPlus (model/orienmask_yolo_fpnplus.py:9-90);
 * variant 1: OrienMaskYOLO (model/orienmask_yolo.py:8-86: one route8 into a 192-channel neck4, no skips). */
int om_model_create_variant(om_model** out, int variant, int num_anchors, int num_classes);
void om_model_destroy(om_model* m);
int om_model_num_layers(const om_model* m);
int om_model_layer_info(const om_model* m, int index, om_layer_info* info);
size_t om_model_weight_floats(const om_model* m);
/* packed_dev: device pointer to the blob described by om_model_layer_info; it must stay alive
 * (and unchanged) for as long as om_forward is called.  dtype: 0 = float32. */
int om_model_load_weights(om_model* m, const void* packed_dev, size_t bytes, int dtype);

/* ---- precision of om_forward's convolutions (no counterpart in the reference, whose convolutions are whatever cuDNN /
 * MKLDNN run) -----------------------------------------------------------------------------------------------------------
 * 0: operands fp32 on v_mfma_f32_32x32x2_f32 (products exact, fp32 accumulate).
 * 1: SPLIT operands, in EVERY convolution: the fused F(4,3) kernel of the stride-1 3x3 layers (conv_wino14.hip), the implicit
 *    GEMM of the 1x1 / stride-2 / head layers (conv_igemm_split.hip), and backbone.conv1 + backbone.conv2.0, which om_forward
 *    runs as ONE kernel (conv_stem2.hip: conv1's 27-term products on the matrix pipe with hi/lo operands too; equal to
 *    om_conv2d_stem followed by om_conv2d_split within 2e-6 of the tensor's scale, not bit for bit).  That fusion is off
 *    while om_model_keep_activations is on, so om_layer_output_view then shows conv1 / conv2.0 from the two-kernel path
 *    (conv1 with fp32 operands); tests/test_hip_parity.py::test_stem2_split_matches_two_kernels compares the two.  Every fp32 operand x
 *    is carried as hi = fp16(x), lo = fp16(x - hi) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with
 *    fp32 accumulation (the lo*lo term, <= 2^-22 of the product, is dropped): 5.3x the matrix rate at the same bytes per
 *    element.  Representation error <= 2^-22 |x| for |x| >= 2^-14 * 2^11 and 2^-25 absolute below (measured end to end in
 *    DESIGN.md 3.5).  Weights are pre-scaled per output channel by the packer; activations are split as they are, so a
 *    layer input must stay below fp16's 65504 in magnitude -- for a stride-1 3x3 layer its TRANSFORMED input, up to 10x
 *    the activation in the fused F(4,3) kernel om_forward runs (20x in om_conv2d_winograd24_split).  An operand beyond that range turns the layer's outputs into NaN, and every split-operand kernel
 *    raises OM_STATUS_SPLIT_RANGE in the forward's status word when it stores a non-finite output (below): the caller
 *    re-runs that batch with mode 0 (orienmask_amd/model.py does).  Needs om_model_load_weights_split; activations between
 *    layers stay fp32.  In this mode the stride-1 3x3 layers run the fused F(4,3)-along-the-rows kernel (conv_wino14.hip:
 *    the transformed input never leaves the CU) at EVERY size (mode 0 switches between F(2x4,3x3) and F(2x2,3x3) at 1700
 *    1/32-scale cells per batch), so an image's outputs do not depend on the batch it is in. */
size_t om_model_weight_split_words(const om_model* m);
int om_model_load_weights_split(om_model* m, const void* packed_split_dev, size_t bytes);
int om_model_set_precision(om_model* m, int mode);
/* Latency mode of precision 1 (no counterpart in the reference, whose published figure is bs = 1 FPS: README.md:5,
 * infer.py:143-172).  cells > 0: a forward whose batch has fewer than `cells` 1/32-scale cells (B * H/32 * W/32; one 544 x 544
 * image has 289) runs its stride-1 3x3 layers as DIRECT convolutions with split operands in the implicit GEMM -- small tiles,
 * one short round -- instead of the fused F(4,3) kernel, whose 128 x 64 tiles leave most of the chip idle at a few images.
 * Same layer, other summation: heads agree with the fused form to ~1e-6 of scale, not bit for bit, so an image's outputs then
 * depend on which side of the switch its batch is.  0 (default): off -- outputs independent of the batch size. */
int om_model_set_latency_cells(om_model* m, long long cells);
/* Latency mode only: the implicit GEMM's launches of few tiles (<= 256 tiles of 64 x 64 / 128 x 64: one image's 1/16- and
 * 1/32-scale layers are 40 .. 152) cut every tile's k loop into up to `max_parts` parts (default 8; as many as two workgroups per
 * CU take in one round, sixteen k-steps each at least), one workgroup per part; the parts' raw accumulators meet in the
 * workspace's partial-tile area and the part that arrives last sums them IN PART ORDER and stores the tile -- no spinning, and
 * the same bits whichever part is last.  At bs = 1 these layers stream their weights (19 MB for a 512 -> 1024 3x3 layer) and
 * the number of CUs that request decides the rate.  1: whole tiles. */
int om_model_set_latency_ksplit(om_model* m, int max_parts);
int om_model_get_precision(const om_model* m);
/* Precision mode 1 only: 1 (default) = the routes and skips (the 1x1 layers whose output the reference up-samples and concatenates,
 * orienmask_yolo_fpnplus.py:78-86) store ONE copy at their own resolution and the 1x1 layer behind the concat reads them
 * up-sampled (om_conv2d_split_gather); 0 = they store their output replicated into the concat buffer, as modes 0 / fp16 do.
 * Same values bit for bit; changes om_forward_workspace_bytes (call it again).  Not active while om_model_keep_activations
 * is on (om_layer_output_view reports the concat slices). */
int om_model_set_upsample_on_read(om_model* m, int enable);

size_t om_forward_workspace_bytes(const om_model* m, int B, int H, int W);
/* x: [B,3,H,W] float32 NCHW, H and W multiples of 32.
 * bbox32/16/8: [B, H/s, W/s, 256] float32 NHWC, channels [0, A*(5+C)) valid  (the caller views
 *              them as [B, A*(5+C), H/s, W/s] with strides)
 * oriens:      [B, 6*A, H/4, W/4] float32 NCHW contiguous */
int om_forward(om_model* m, const float* x, int B, int H, int W, float* bbox32, float* bbox16, float* bbox8,
               float* oriens, void* workspace, size_t ws_bytes, om_stream stream);
/* The forward's STATUS WORD: one int32 inside the workspace handed to om_forward, om_forward_status_offset() bytes from its
 * start, cleared when a forward starts and OR-ed by its kernels (om_forward_f16 has no conditions to report).  The caller reads it with whatever device-to-host copy it does anyway after the step
 * (orienmask_amd/eval.py reads it together with the detection counts).  0 = the outputs are valid. */
#define OM_STATUS_SPLIT_RANGE 1 /* precision mode 1: a split-operand layer stored a non-finite output -- an operand left the
                                   fp16 range of the hi/lo representation (or the fp32 result itself is non-finite): re-run
                                   the batch with om_model_set_precision(m, 0) */
#define OM_STATUS_SK_TIMEOUT 2  /* a stream-K finisher gave up waiting for its partner workgroup's partial tile (bounded spin
                                   so that a fault cannot hang the device): the outputs are invalid */
size_t om_forward_status_offset(const om_model* m, int B, int H, int W);

/* ---- fp16 activations, fp32 accumulate (BASELINE.json configs[4]; no reduced-precision path exists in the
 * reference -- the arithmetic is defined by oracle/orienmask_ref.py:forward_f16) -------------------------------
 * Activations between layers and the convolution weights are IEEE fp16; sums, BatchNorm scale/shift, LeakyReLU and the
 * residual add are fp32, rounded once per layer at the store; the stem reads the fp32 image; the four head tensors are
 * written fp32 exactly as om_forward writes them.  Needs BOTH blobs: om_model_load_weights (scale/shift, stem) and
 * om_model_load_weights_f16 (the fp16 weights laid out per om_layer_info.w16_off, om_model_weight_halfs() halfs). */
size_t om_model_weight_halfs(const om_model* m);
int om_model_load_weights_f16(om_model* m, const void* packed_f16_dev, size_t bytes);
size_t om_forward_f16_workspace_bytes(const om_model* m, int B, int H, int W);
int om_forward_f16(om_model* m, const float* x, int B, int H, int W, float* bbox32, float* bbox16, float* bbox8,
                   float* oriens, void* workspace, size_t ws_bytes, om_stream stream);
/* kernel that runs layer `index` in the fp16 configuration (6 = conv3x3_f16_tall_kernel<512,128>, 7 = conv_stem2_f16_kernel: the first
 * two layers in one launch, 8 = a layer computed inside the previous layer's kernel): algo 0 = conv_stem_kernel<f16>, 1 = conv_igemm_f16_kernel<bm,bn>,
 * 4 = conv3x3_f16_kernel<bm,bn> (stride-1 3x3: the three taps of a kernel row share one LDS copy of the input rows) */
int om_layer_tile_f16(const om_model* m, int index, int B, int H, int W, int* bm, int* bn, int* algo);

/* ---- intermediate activations (tests / debugging; BASELINE configs[1] checks the DarkNet-53 features) -------------------
 * Where layer `index`'s output lives inside the workspace of the last om_forward (f16 = 0) / om_forward_f16 (f16 = 1) call
 * at this problem size: an NHWC view [B, H/div, W/div, channels] starting byte_offset bytes into the workspace, pixel
 * stride pix_stride ELEMENTS (a concat buffer's slice has pix_stride > channels).  The four head layers write
 * caller-owned tensors and are rejected; so is a model without om_model_keep_activations(m, 1). */
int om_layer_output_view(const om_model* m, int index, int B, int H, int W, int f16, size_t* byte_offset, int* channels,
                         int* pix_stride, int* div);
/* The forward's activations (and the Winograd scratch) share workspace memory by live range: a tensor's slab is reused once its
 * last consumer has run.  keep = 1 gives every tensor its own slab again (larger om_forward_workspace_bytes) so that
 * om_layer_output_view can be read after the forward; it must be set before the workspace is sized. */
int om_model_keep_activations(om_model* m, int keep);

/* ---- measurement: per-layer durations with HIP events on the stream om_forward launches on
 * (the reference measures with torch.cuda.Event pairs, utils/timer.py:70-82).  While enabled, every
 * om_forward records events around every kernel of every layer; om_profile_read synchronises on them and
 * returns, per layer (graph order of om_model_layer_info) and summed over the recorded forwards, the
 * milliseconds of the layer's main kernel (layer_ms) and of its pre-pass (layer_pre_ms: the Winograd input
 * transform; 0 for single-kernel layers), plus the number of forwards. */
/* Tile shape (rows x channels per workgroup) of the conv kernel instantiation that runs layer `index` at
 * this problem size; 0 x 0 for the stem kernel.  Lets a profile be grouped by kernel.  algo: 0 conv_stem_kernel, 1 conv_igemm_f32,
 * 2 / 3 Winograd F(2x2) GEMM / fused, 5 / 6 F(2x4) GEMM with fp32 / split operands, 7 conv_igemm_split, 8 wino14_split (fused
 * F(4,3)), 9 conv_stem2_split_kernel (precision mode 1: backbone.conv1 + backbone.conv2.0 in one kernel, reported for conv1),
 * 10 "part of the previous layer's kernel" (reported for conv2.0 then: om_forward launches nothing for it), 11 conv_igemm_split in
 * its GATHER form (precision mode 1: the 1x1 layer behind an up-sample + concat reads the low-resolution slices where their
 * producers stored them, om_conv2d_split_gather). */
int om_layer_tile(const om_model* m, int index, int B, int H, int W, int* bm, int* bn, int* algo);
/* algo: 0 = conv_stem_kernel, 1 = conv_igemm_f32_kernel<bm,bn>, 2 = wino_input_kernel + wino_gemm_kernel<bm,bn>,
 *       3 = wino_fused_kernel<bn> (input transform fused into the GEMM's loader),
 *       5 = wino24_input_kernel + wino24_gemm_kernel (Winograd F(2x4,3x3), 64x64 tile) */
int om_profile_enable(om_model* m, int enable);
/* record events only around the layers with layer_mask[i] != 0 (an event pair costs a few microseconds of GPU time, so a
 * timed region that only needs one kernel's durations should not pay for all ~90 layers); om_profile_read then returns 0
 * for the other layers */
int om_profile_enable_layers(om_model* m, const unsigned char* layer_mask, int n_layers);
int om_profile_read(om_model* m, float* layer_ms, float* layer_pre_ms, int n_layers, int* n_forwards);

/* ---- one convolution (unit-test entries) ---------------------------------------------------
 * These entries exist so that parity tests can check single layers; unlike the hot path they own a few device words (the
 * tile-queue ticket, hipMalloc'ed on first use and never freed), which om_forward carves from the caller's workspace. */
/* in: [B,H,W,cin] NHWC (pixel stride in_pix_stride floats); w/scale/shift as in om_layer_info;
 * res: optional [B,Ho,Wo,cout] NHWC added after the activation; out: [B,Ho,Wo,cout] NHWC. */
int om_conv2d(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* w,
              const float* scale, const float* shift, int cout, int ksize, int stride, int leaky,
              const float* res, int res_pix_stride, float* out, int out_pix_stride, om_stream stream);
/* The same with the output layouts om_forward uses: out_mode 0 = NHWC; 1 = NHWC with every output pixel replicated up x up
 * (NearestUpsample fused into the producer, model/base.py:95-101: out is [B, Ho*up, Wo*up, ...]); 2 = NCHW contiguous
 * [B, cout, Ho, Wo] (the orientation head). */
int om_conv2d_mode(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* w,
                   const float* scale, const float* shift, int cout, int ksize, int stride, int leaky,
                   const float* res, int res_pix_stride, float* out, int out_pix_stride, int out_mode, int up,
                   om_stream stream);
/* The same layer through Winograd F(2x2,3x3) (3x3, stride 1): u = G g G^T as [16][cout_pad][cin];
 * scratch: om_conv2d_winograd_scratch_bytes(B,H,W,cin) bytes of device memory for the transformed input. */
size_t om_conv2d_winograd_scratch_bytes(int B, int H, int W, int cin);
int om_conv2d_winograd(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* u,
                       const float* scale, const float* shift, int cout, int leaky, const float* res,
                       int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                       om_stream stream);
/* fp16 layer (unit-test entry): in/res/w fp16 (w laid out as om_layer_info.w16_off describes), scale/shift fp32,
 * out fp16 NHWC, or fp32 NHWC when out_f32 = 1; pixel strides in elements of the respective type. */
int om_conv2d_f16(const void* in, int B, int H, int W, int cin, int in_pix_stride, const void* w, const float* scale,
                  const float* shift, int cout, int ksize, int stride, int leaky, const void* res, int res_pix_stride,
                  void* out, int out_pix_stride, int out_f32, om_stream stream);
int om_conv2d_stem_f16(const float* in, int B, int H, int W, const float* w, const float* scale, const float* shift,
                       int cout, void* out, om_stream stream);
/* The same layer through Winograd F(2x4,3x3): u = G_y g G_x^T as [24][cout_pad][cin], cout_pad = cout rounded up to 64. */
size_t om_conv2d_winograd24_scratch_bytes(int B, int H, int W, int cin);
int om_conv2d_winograd24(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* u,
                         const float* scale, const float* shift, int cout, int leaky, const float* res,
                         int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                         om_stream stream);
/* one ConvBNRelu with split operands (conv_igemm_split.hip): w_split / scale_split as om_layer_info.wsplit_off /
 * wsplit_scale_off describe for a layer without F(2x4) form; other arguments as om_conv2d_mode.  tile_bm x tile_bn: 0, 0 =
 * the tile shape om_forward would pick, else one of the built shapes 256x128, 128x128, 128x64, 64x64, 128x32 with tile_bn
 * dividing cout_pad (per call: tile sweeps and tests of every instantiation).  status_dev: optional device int32 that the
 * kernel ORs OM_STATUS_* bits into (the caller clears it), NULL = not reported. */
int om_conv2d_split(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* w_split,
                    const float* scale_split, const float* shift, int cout, int ksize, int stride, int leaky, const float* res,
                    int res_pix_stride, float* out, int out_pix_stride, int out_mode, int up, int tile_bm, int tile_bn,
                    int32_t* status_dev, om_stream stream);
/* om_conv2d_split with a tile's k loop cut into up to max_parts (1 .. 8) parts, as the latency mode's small launches run
 * (om_model_set_latency_ksplit: fewer where the tile count or the k loop's length caps it).  Takes effect for the 128x64 and
 * 64x64 shapes of at most 256 tiles; anything else runs as om_conv2d_split.  The partial tiles live in library-owned memory
 * (unit-test entry; om_forward uses the caller's workspace). */
int om_conv2d_split_k(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* w_split,
                      const float* scale_split, const float* shift, int cout, int ksize, int stride, int leaky, const float* res,
                      int res_pix_stride, float* out, int out_pix_stride, int out_mode, int up, int tile_bm, int tile_bn,
                      int max_parts, int32_t* status_dev, om_stream stream);
/* The 1x1 layer behind an up-sample + concat, as om_forward runs neck16.0 / neck8.0 / neck4.0 in precision mode 1
 * (conv_igemm_split.hip, GATHER form): the input channels are the concatenation of nseg (1..4) NHWC tensors, segment g with
 * seg_channels[g] channels (a multiple of 32) stored at [B, H/seg_up[g], W/seg_up[g], seg_pix_stride[g]] (seg_up a power of two
 * dividing H and W) and read nearest-up-sampled; cout_pad must be a multiple of 128.  Bit-identical to om_conv2d_split over
 * the materialised concat.  Replaces F.interpolate(scale_factor, 'nearest') + torch.cat + the following 1x1 Conv-BN-LeakyReLU
 * of /root/reference/model/orienmask_yolo_fpnplus.py:78-86 (forward: route / skip concatenations). */
int om_conv2d_split_gather(int nseg, const float* const* seg_ptr, const int* seg_channels, const int* seg_pix_stride,
                           const int* seg_up, int B, int H, int W, const void* w_split, const float* scale_split, const float* shift,
                           int cout, int leaky, float* out, int out_pix_stride, int32_t* status_dev, om_stream stream);
/* ... with split operands (om_model_set_precision mode 1): u_split as om_layer_info.wsplit_off describes, scale_split =
 * scale * 2^-e per output channel; scratch as for om_conv2d_winograd24; status_dev as for om_conv2d_split. */
int om_conv2d_winograd24_split(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* u_split,
                               const float* scale_split, const float* shift, int cout, int leaky, const float* res,
                               int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                               int32_t* status_dev, om_stream stream);
/* The stride-1 3x3 layer in the fused split-operand form om_forward runs in precision mode 1 (conv_wino14.hip: Winograd F(4,3)
 * along the rows, the input transform inside the kernel, no scratch): u14_split = the packed weights
 * [cout_pad/64][cin/16][6][3][64][32 halfs] (orienmask_amd/pack.py: winograd14_weights_split), cout_pad = cout rounded up to 64,
 * cin a multiple of 16; scale_split = scale * 2^-e; status_dev as for om_conv2d_split. */
int om_conv2d_wino14_split(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* u14_split,
                           const float* scale_split, const float* shift, int cout, int leaky, const float* res,
                           int res_pix_stride, float* out, int out_pix_stride, int32_t* status_dev, om_stream stream);
/* The same layer as TWO kernels with a 128 x 128 tile (round 6; conv_wino14.hip: wino14_v_kernel writes the transformed input
 * V = [cin/16][6][B (H + 2)][ceil(W / 4)] entries of 64 bytes into `scratch`, wino14_wide_kernel reads it by LDS-DMA: eight waves of
 * 64 entries x 32 channels x six planes, the fused kernel's epilogue): the same products in the same order, BIT-IDENTICAL outputs.
 * cout_pad must be a multiple of 128 and cout of 4, the views 16-byte aligned.  om_forward (precision mode 1) runs the layers with
 * at least 512 input channels this way -- where 2.5 x the input through HBM for the pre-pass is small next to the layer's work --
 * unless om_set_wino14_wide(0) / OM_NO_W14_WIDE=1 (process-wide A/B switch; default on). */
size_t om_conv2d_wino14_wide_scratch_bytes(int B, int H, int W, int cin);
int om_conv2d_wino14_wide(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* u14_split,
                          const float* scale_split, const float* shift, int cout, int leaky, const float* res,
                          int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                          int32_t* status_dev, om_stream stream);
int om_set_wino14_wide(int on);
int om_get_wino14_wide(void);
/* Which kernel runs that layer (process-wide; the outputs are bit-identical): 0 (default; environment OM_W14_VARIANT) the
 * twelve-wave kernel (conv_wino14.hip) everywhere; 1 the four-dual-role-wave kernel of round 5 (conv_wino14d.hip: one wave per
 * SIMD, accumulators owned by name) wherever it applies -- an even number >= 2 of 16-channel chunks, 16-byte aligned views.
 * The second form is kept as a measured alternative (8-25 % slower: profiles/r05_experiments.md), not as the product's path:
 * only a library built with `make -C orienmask_amd/csrc W14D=1` contains it (om_wino14_dual_built() == 1); in the default
 * library om_set_wino14_variant(1) returns OM_EINVAL. */
int om_set_wino14_variant(int variant);
int om_get_wino14_variant(void);
int om_wino14_dual_built(void);
/* A/B switches of the two first-layers fusions (process-wide; default on; OM_NO_STEM3=1 / OM_NO_STEM2_F16=1 in the environment
 * turn them off before the first use): which = 0 the third layer (backbone.conv2.1.conv.0) inside the split-operand
 * first-two-layers kernel, which = 1 the fp16 first-two-layers kernel.  Off -> the separate kernels: bit-identical results for
 * which = 0; for which = 1 conv1's fp32 sums are formed in another order before their one rounding to fp16 (a few per cent of the
 * activations move by one fp16 step: inside the fp16 configuration's tolerance, not bit-identical).  A
 * view the fused launcher cannot take (alignment, pixel stride, descriptor size) runs the separate kernels by itself. */
int om_set_stem_fusion(int which, int on);
int om_get_stem_fusion(int which);      /* 1 on, 0 off, -1 bad argument */
/* Which kernel runs the stride-1 3x3 layers of the fp16 configuration (process-wide; the results are the same sums in the same
 * order, bit-identical): 0 the 256 x 128 / 128 x 128 / 128 x 64 shared-patch kernel everywhere (rounds 1-4); 1 (default;
 * environment OM_C3_TALL) the tall-patch kernel of round 5 (conv3x3_f16.hip: 512 raster pixels x 128 channels per eight-wave
 * workgroup, one input patch per 32-channel chunk for all nine taps) where the tile chooser picks it; 2 wherever it can run
 * (cout_pad a multiple of 128, rows of at most 191 pixels). */
int om_set_conv3x3_f16_variant(int mode);
int om_get_conv3x3_f16_variant(void);
/* first layer: in [B,3,H,W] NCHW -> out [B,H,W,cout] NHWC, 3x3 stride 1. */
int om_conv2d_stem(const float* in, int B, int H, int W, const float* w, const float* scale,
                   const float* shift, int cout, float* out, om_stream stream);
/* The first TWO layers as om_forward runs them in precision mode 1 (conv_stem2.hip): backbone.conv1 (3 -> 32, 3x3, BN, LeakyReLU;
 * w1 [32][27], scale1 / shift1 [32] as for om_conv2d_stem) and backbone.conv2.0 (32 -> 64, 3x3 stride 2, BN, LeakyReLU when
 * leaky2; w2_split / scale2_split as om_layer_info.wsplit_off / wsplit_scale_off describe, shift2 [64]) in one kernel:
 * in [B,3,H,W] NCHW (H, W even) -> out [B,H/2,W/2,out_pix_stride] NHWC.  conv1's products run on the matrix pipe with split
 * operands here (fp32 multiply-adds in om_conv2d_stem): the same values as om_conv2d_stem followed by om_conv2d_split to 2e-6 of
 * the tensor's scale, not bit for bit.  status_dev as for om_conv2d_split.
 * Replaces /root/reference/model/backbone/darknet.py:20-22 (conv1, conv2's first block) as called from model/base.py:104-137. */
int om_conv2d_stem2_split(const float* in, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                          const void* w2_split, const float* scale2_split, const float* shift2, int cout2, int leaky2, float* out,
                          int out_pix_stride, int32_t* status_dev, om_stream stream);
/* ... with the layer behind them in the same launch (round 5): backbone.conv2.1.conv.0, the 64 -> 32 1x1 convolution of the first
 * residual block (/root/reference/model/backbone/darknet.py:9-13), computed on each tile's conv2.0 outputs while they are in the
 * workgroup: `out` is written as by om_conv2d_stem2_split (the block's residual reads it), `out3` [B,H/2,W/2,out3_pix_stride] gets
 * what om_conv2d_split would compute from it -- the same split8 / three-instruction arithmetic per 16 channels, bit for bit.
 * w3_split / scale3_split: conv_weights_split rows [32][4][4][8] halfs and scale * 2^-e; cout3 = 32. */
int om_conv2d_stem3_split(const float* in, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                          const void* w2_split, const float* scale2_split, const float* shift2, int cout2, int leaky2, float* out,
                          int out_pix_stride, const void* w3_split, const float* scale3_split, const float* shift3, int cout3, int leaky3,
                          float* out3, int out3_pix_stride, int32_t* status_dev, om_stream stream);
/* The same two layers as om_forward_f16 runs them (conv_stem2.hip: conv_stem2_f16_kernel; round 5): conv1 from the fp32 image with
 * fp32 weights (fp32-level sums, one rounding to fp16 -- the activation om_conv2d_stem_f16 stores, never written here), conv2.0 on
 * fp16 operands: w2_f16 = its fp16 rows [64][9 * 32] (om_layer_info.w16_off), scale2 / shift2 [64] fp32; out [B,H/2,W/2,out_pix_stride]
 * fp16 NHWC, 8-byte aligned.  Equal to om_conv2d_stem_f16 followed by om_conv2d_f16 up to conv1's rounding ties. */
int om_conv2d_stem2_f16(const float* in, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                        const void* w2_f16, const float* scale2, const float* shift2, int cout2, int leaky2, void* out,
                        int out_pix_stride, om_stream stream);

/* ---- preprocess: FastCOCOTransform.__call__ + pad, fused ------------------------------
 * Replaces /root/reference/data/transform.py:444-510 (permute, Resize = F.interpolate bilinear
 * align_corners=False, Normalize = (x - mean) / std) and /root/reference/infer.py:21-32 (zero padding to a
 * multiple of 32).  in: [N,h,w,3] float32 HWC; the image is resized to resize_h x resize_w, normalised and
 * placed at (pad_top, pad_left) of the [N,3,out_h,out_w] NCHW output; everything else is pad_value. */
int om_preprocess(const float* in_nhwc, int N, int h, int w, int resize_h, int resize_w, const float* mean3,
                  const float* std3, int pad_top, int pad_left, int out_h, int out_w, float pad_value, float* out_nchw,
                  om_stream stream);
/* pad() alone on an NCHW tensor of `planes` = N*C planes. */
int om_pad_nchw(const float* in, long long planes, int h, int w, int pad_top, int pad_left, int out_h, int out_w,
                float pad_value, float* out, om_stream stream);

/* ---- postprocess -------------------------------------------------------------------------- */
size_t om_postprocess_workspace_bytes(const om_post_cfg* cfg, int B);
/* Limits of the fused path (the reference has none): nms_pre <= 1024; 1..3 scales of 1..3 anchors; conf_thresh >= 0 (the
 * decode skips a candidate whose sigmoid(objectness) is not above it before reading its class logits); num_classes such that
 * (2047 / C + 2) * max(C & ~31, C & 31) + 63 <= 6144 -- every C < 2048 (a decode thread's visits per sweep are unrolled: 10 of
 * them cover C <= 251, a second instantiation with 24 the rest, e.g. LVIS's 1203; csrc/post.hip DEC_VISITS / DEC_VISITS_MANY);
 * larger class counts return OM_EINVAL with that message.
 * out_bbox [B,nms_post,5] (cx,cy,w,h normalised, score); out_cls [B,nms_post] int64;
 * out_mask [B,nms_post,image_h,image_w] uint8 0/1 (rows >= out_count[b] are left untouched);
 * out_count [B] int32; out_keep [B,nms_post] int32 position of each detection in the
 * pre-NMS candidate list, may be NULL.  oriens as written by om_forward. */
int om_postprocess(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8,
                   const float* oriens, int B, float* out_bbox, int64_t* out_cls, uint8_t* out_mask,
                   int32_t* out_count, int32_t* out_keep, void* workspace, size_t ws_bytes, om_stream stream);
/* om_postprocess in its two halves (it IS detect followed by assemble on one stream).  detect: decode, threshold, top-nms_pre,
 * batched NMS, top-nms_post (postprocess.py:102-154) -- reads the box heads only; writes out_bbox / out_cls / out_count /
 * out_keep and the detections' mask constants into the workspace.  assemble: the masks (postprocess.py:69-72,141-144,156-164) of
 * those detections from the orientation head, with the same workspace.  Apart they let the first half run on another stream
 * while the forward is still computing the orientation branch: om_model_attach_postprocess. */
int om_postprocess_detect(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8, int B,
                          float* out_bbox, int64_t* out_cls, int32_t* out_count, int32_t* out_keep, void* workspace, size_t ws_bytes,
                          om_stream stream);
int om_postprocess_assemble(const om_post_cfg* cfg, const float* oriens, int B, const int32_t* out_count, uint8_t* out_mask,
                            void* workspace, size_t ws_bytes, om_stream stream);
/* The step as ONE call sequence on the forward (tester.py:39-44's two lines; infer.py:154-156).  While a postprocess is
 * attached, om_forward (fp32 / split precision) also runs it into the attached buffers: decode + select are launched on a second
 * stream that the library creates, forked off the caller's stream by an event as soon as the last box-head layer is launched and
 * joined behind the forward's last layer -- they read the box heads only, the select kernel is ONE workgroup per image, so beside
 * the skips, neck4 and the orientation head they cost nothing -- and the mask kernel follows on the caller's stream.  On return
 * everything is ordered before later work on the caller's stream; under stream capture the second stream joins the capture
 * through the same events.  Same kernels on the same inputs as om_forward followed by om_postprocess: same bits.  cfg == NULL
 * detaches.  The buffers must stay valid while attached (orienmask_amd/eval.py attaches around one forward). */
int om_model_attach_postprocess(om_model* m, const om_post_cfg* cfg, float* out_bbox, int64_t* out_cls, uint8_t* out_mask,
                                int32_t* out_count, int32_t* out_keep, void* post_workspace, size_t post_ws_bytes);

/* The same postprocess around a caller-supplied NMS callable -- the reference takes ANY nms_func(dets, cls) -> (dets[keep],
 * cls[keep], keep) (/root/reference/eval/orienmask_yolo_postprocess.py:9-11,146-148); the fused om_postprocess implements
 * batched_nms itself.  om_postprocess_candidates: decode, threshold and top-nms_pre (postprocess.py:102-122): cand_dets
 * [B,nms_pre,5] (cx, cy, w, h, score), cand_cls [B,nms_pre], cand_field [B,nms_pre] (anchor field of the candidate: its
 * orientation channels are 2 * field, 2 * field + 1), cand_count [B], in the order the reference hands them to self.nms.  The
 * caller runs its callable per image, applies the top-nms_post of postprocess.py:150-154 and hands the survivors to
 * om_postprocess_masks: dets [B,nms_post,5], field [B,nms_post], count [B] (device) -> out_mask [B,nms_post,image_h,image_w]
 * (postprocess.py:156-164).  Same workspace size as om_postprocess. */
int om_postprocess_candidates(const om_post_cfg* cfg, const float* bbox32, const float* bbox16, const float* bbox8, int B,
                              float* cand_dets, int64_t* cand_cls, int32_t* cand_field, int32_t* cand_count, void* workspace,
                              size_t ws_bytes, om_stream stream);
int om_postprocess_masks(const om_post_cfg* cfg, const float* oriens, int B, const float* dets, const int32_t* field,
                         const int32_t* count, uint8_t* out_mask, void* workspace, size_t ws_bytes, om_stream stream);

/* measurement: launch geometry and resource use of the three postprocess kernels (which: 0 = post_decode_kernel,
 * 1 = post_select_kernel, 2 = post_mask_kernel) as the runtime reports them for this device: threads per workgroup, VGPRs
 * per lane, static LDS bytes per workgroup and the number of workgroups one CU can hold (hipOccupancyMaxActiveBlocks...).
 * bench.py turns them into waves/SIMD against the gfx950 limit of 8 (north_star: "occupancy for NMS/mask-assembly"). */
int om_post_kernel_occupancy(int which, int* threads, int* vgprs, int* lds_bytes, int* max_blocks_per_cu);

/* ---- COCO-format conversion ------------------------------------------------------------
 * om_recover_bbox: COCOMetrics._recover_shape_bbox, /root/reference/eval/coco_eval.py:146-189.
 *   bbox [K,stride] normalised (cx,cy,w,h,..) -> out_xywh [K,4] top-left x, y, w, h in original-image pixels.
 *   collate_pad6 = (left,right,top,down,h,w) and pad6 = (top,down,left,right,h,w) are HOST arrays or NULL.
 * om_recover_masks_rle: COCOMetrics._recover_shape_segm (:191-205: crop the paddings, flips, bilinear resize
 *   to the original size, round) + the column-major run lengths of pycocotools' rleEncode (:120-122).
 *   mask [K,H,W] u8 0/1; crop_* = total padding to strip on each side; counts [K][max_runs] u32 receives the
 *   run lengths (first run counts zeros); n_runs [K] their number (> max_runs: buffer too small, counts of
 *   that mask are undefined); resized_or_null: optional [K,orig_h,orig_w] u8 copy of the resized masks. */
int om_recover_bbox(const float* bbox, int K, int stride, const int32_t* collate_pad6, const int32_t* pad6, int hflip,
                    int vflip, int orig_h, int orig_w, float* out_xywh, om_stream stream);
int om_recover_masks_rle(const uint8_t* mask, int K, int H, int W, int crop_top, int crop_down, int crop_left,
                         int crop_right, int hflip, int vflip, int orig_h, int orig_w, uint32_t* counts, int max_runs,
                         int32_t* n_runs, uint8_t* resized_or_null, om_stream stream);
/* om_recover_masks_rle_strings: the same conversion for a whole BATCH in one launch, ending on the device with what the
 *   reference gets from maskUtils.encode(...)['counts'] (/root/reference/eval/coco_eval.py:120-122): pycocotools' rleToString
 *   of every mask's run lengths (published algorithm of cocoapi common/maskApi.c; the library is absent offline, so the
 *   STRING is pinned against oracle/rle_ref.c only).  images: HOST array, one entry per image (K may be 0); masks are numbered
 *   image after image.  counts [sum K][max_runs] scratch for the run lengths, n_runs [sum K] (> max_runs: that mask needs a
 *   larger buffer, its str_off is -1).  Every mask appends its string to ONE byte buffer: str_cursor[0] = bytes reserved
 *   (zeroed by the caller), str_cursor[1] = masks whose string did not fit str_capacity; str_off / str_len [sum K].  The host
 *   then copies the cursor + offsets and bytes[0:cursor] once per batch instead of 30 MB of masks per image. */
typedef struct om_rle_image {
    const uint8_t* mask;        /* [K,H,W] u8 0/1, device */
    int32_t K, H, W;
    int32_t crop_top, crop_down, crop_left, crop_right;
    int32_t hflip, vflip, orig_h, orig_w;
} om_rle_image;
int om_recover_masks_rle_strings(const om_rle_image* images, int n_images, uint32_t* counts, int max_runs, int32_t* n_runs,
                                 uint8_t* str_bytes, long long str_capacity, int32_t* str_cursor, int32_t* str_off,
                                 int32_t* str_len, om_stream stream);

/* ---- unit-test entry: the elementary functions the decode uses, restated bit-exactly from what torch-CPU runs at the
 * reference's call sites eval/orienmask_yolo_postprocess.py:127-136 (csrc/ref_math.h).  func: 0 = glibc expf (torch's
 * scalar loop), 1 = Sleef expf_u10 (torch's vectorised loop), 2 / 3 = sigmoid through either, 4 = sigmoid of a
 * [rows][num_classes] array exactly as predict[..., 5:].sigmoid() evaluates it (classes below (C/32)*32 vectorised, the
 * row tail scalar), 5 = correctly rounded expf, 6 = the mask predicate's building block (post.hip: abs_minus_bits): y[i] = the IEEE
 * difference |x[i]| - x[i ^ 1] (n even), whose SIGN decides |d| < t in post_mask_kernel -- the test checks the signs of the
 * infinite and NaN cases on the device. */
int om_ref_math(const float* x, long long n, int func, int num_classes, float* y, om_stream stream);

/* ---- NMS: the reference's native export nms(dets[n,5], threshold) -> keep (eval/src/nms_cpu.cpp:65-75,
 *      eval/src/nms_cuda.cpp:8-17), n <= 65536 (0 from om_nms_workspace_bytes above that).
 *      om_nms = om_nms_ex(semantics 0): the CPU backend (IoU >= thresh suppresses, areas from the corners cx +- w/2, keep
 *      in ascending input order, nms_cpu.cpp:4-63).  semantics 1: the CUDA backend (IoU > thresh, areas w*h, keep in
 *      score-descending order, nms_kernel.cu:13-140); score ties, which torch's CUDA sort leaves unspecified, are visited
 *      in ascending input order. */
size_t om_nms_workspace_bytes(int n);
int om_nms(const float* dets, int n, float thresh, int64_t* keep, int32_t* n_keep, void* workspace,
           size_t ws_bytes, om_stream stream);
int om_nms_ex(const float* dets, int n, float thresh, int semantics, int64_t* keep, int32_t* n_keep, void* workspace,
              size_t ws_bytes, om_stream stream);

/* ---- Several batches in flight.  Every entry point only enqueues kernels on the caller's stream and keeps no per-call state in
 *      the model handle (profiling apart): om_forward / om_forward_f16 / om_postprocess may be issued for different batches
 *      on different HIP streams at the same time, provided each batch in flight has its OWN workspace (and output buffers);
 *      the weights are only read.  Kernels of the streams then share compute units, and results do not depend on what the
 *      other stream runs (tests/test_hip_parity.py::test_postprocess_and_forward_are_stable_beside_other_streams).  The host
 *      side of this is orienmask_amd/pipeline.py (InFlightPipeline): +13 % fp32 / +19 % fp16 images/s at 32 x 544 x 544. */

#ifdef __cplusplus
}
#endif
#endif /* ORIENMASK_HIP_H */
