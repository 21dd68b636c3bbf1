// Host side of liborienmask_hip.so: the OrienMaskYOLOFPNPlus inference graph as a list of fused
// convolution launches over NHWC buffers carved from one caller-provided workspace.
//
// The graph restates /root/reference/model/orienmask_yolo_fpnplus.py:9-90 and
// /root/reference/model/backbone/darknet.py:18-54 (built here from the channel/stride rules,
// not translated): DarkNet-53 stages 1/2/8/8/4, four 5-conv necks, two up-sampling routes,
// three 2-conv box heads, four skip projections and the 6-conv orientation head.
//
// MI355X-first decisions:
//   * every tensor is NHWC; torch.cat never runs: each concat is ONE buffer and its producers
//     write their channel slice (route/skip outputs are written already nearest-upsampled,
//     reference model/base.py:95-101); consumers read strided views;
//   * residual adds, BatchNorm and LeakyReLU live in the conv epilogue (conv_igemm.hip);
//   * the three box heads are written NHWC with a 256-float pixel stride (what the decode kernel
//     wants), the orientation head NCHW (what the mask kernel wants);
//   * no allocation, no synchronisation: ~90 launches on the caller's stream.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "om_common.h"

namespace om {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

enum : int { BUF_INPUT = -1, BUF_BBOX32 = -2, BUF_BBOX16 = -3, BUF_BBOX8 = -4, BUF_ORIENS = -5 };
constexpr int HEAD_PIX_STRIDE = 256;

struct BufDef {
    int div;    // spatial size = image / div
    int C;      // floats per pixel
};

struct View {
    int buf;
    int ch_off;
};

struct LayerDef {
    om_layer_info info;
    View in, out, res;
    bool has_res = false;
    bool stem = false;
    int in_div = 1;      // spatial divisor of the input
    int out_mode = 0, up = 1;
    // Up-sampling on read (split-operand mode, find_gathers()): an up-sampling producer (out_mode 1) also owns `side`, a buffer at
    // its OWN resolution; the 1x1 layer that reads the concat buffer lists, in channel order, where each slice really is.
    int side = -1;
    struct Seg { int producer; View view; int channels, up; };      // producer: layer index whose `side` holds the slice, or -1: `view`
    std::vector<Seg> gather;
};

}  // namespace om

// process-wide A/B switch (default on; om_set_wino14_wide, or OM_NO_W14_WIDE=1 in the environment read once): om_forward runs the
// stride-1 3x3 layers with at least 512 input channels of precision mode 1 in the two-kernel wide form (conv_wino14.hip:
// wino14_v_kernel + wino14_wide_kernel); off: the fused kernel everywhere.  Bit-identical either way.
static int g_wino14_wide = -1;
static bool wino14_wide_on() {
    if (g_wino14_wide < 0) {
        const char* e = std::getenv("OM_NO_W14_WIDE");
        g_wino14_wide = (e && e[0] == '1') ? 0 : 1;
    }
    return g_wino14_wide != 0;
}

struct om_model {
    int num_anchors = 0, num_classes = 0;
    int variant = 0;     // 0: OrienMaskYOLOFPNPlus, 1: OrienMaskYOLO (single route8 into a 192-channel neck4)
    std::vector<om::BufDef> bufs;
    std::vector<om::LayerDef> layers;
    size_t weight_floats = 0;
    const float* weights = nullptr;
    size_t weight_halfs = 0;             // fp16 copy of the convolution weights (om_model_load_weights_f16)
    const _Float16* weights16 = nullptr;
    size_t split_words = 0;              // hi/lo fp16 pairs of the F(2x4) weights + their scales (om_model_load_weights_split)
    const float* weights_split = nullptr;
    int precision = 0;                   // 0: fp32 operands, 1: split operands in the F(2x4) GEMMs
    // optional per-layer timing with HIP events on the launch stream (om_profile_*)
    bool profiling = false;
    std::vector<unsigned char> prof_mask;    // empty: every layer; else 1 = record events for this layer
    std::vector<hipEvent_t> ev_pool;     // 3 events per (recorded forward, layer): start, mid, stop
    size_t ev_used = 0;
    int prof_forwards = 0;

    int new_buf(int div, int C) {
        bufs.push_back({div, C});
        return (int)bufs.size() - 1;
    }
    int pix_stride(int buf) const { return buf >= 0 ? bufs[buf].C : om::HEAD_PIX_STRIDE; }

    // Appends one convolution and reserves its slice of the weight blob.
    void add(const std::string& name, int cin, int cout, int ks, int stride, bool bn, om::View in, int in_div,
             om::View out, const om::View* res = nullptr, int out_mode = 0, int up = 1, bool stem = false) {
        om::LayerDef L;
        std::memset(&L.info, 0, sizeof(L.info));
        std::snprintf(L.info.name, sizeof(L.info.name), "%s", name.c_str());
        L.info.cin = cin; L.info.cout = cout; L.info.cout_pad = om::round_up(cout, 32);
        L.info.ksize = ks; L.info.stride = stride; L.info.has_bn = bn ? 1 : 0; L.info.leaky = bn ? 1 : 0;
        L.info.w_off = (int64_t)weight_floats;
        weight_floats += (size_t)L.info.cout_pad * ks * ks * cin;
        weight_floats = om::align_up(weight_floats, 4);
        L.info.scale_off = (int64_t)weight_floats; weight_floats += L.info.cout_pad;
        L.info.shift_off = (int64_t)weight_floats; weight_floats += L.info.cout_pad;
        L.info.wino_off = -1;
        L.info.wino_alt_off = -1;
        L.info.wino_planes = 0;
        if (ks == 3 && stride == 1 && !stem && L.info.cout_pad % 64 == 0 && cin % 32 == 0) {
            // F(2x4,3x3) (24 planes) at every scale.  At 1/32 scale (17 x 17 at 544: 18 % of a 2 x 4 tiling is padding, 368 tiles for 512
            // resident workgroups) a layer takes the same time as with F(2x2) when it runs alone (0.441 vs 0.443 ms at bs=32) but
            // leaves a quarter of the chip to the other batch in flight: +0.9 % end to end with two in flight, -0.5 % one at a time
            // (same-box A/B, profiles/r02_experiments.md).
            L.info.wino_planes = 24;
            weight_floats = om::align_up(weight_floats, 4);
            L.info.wino_off = (int64_t)weight_floats;
            weight_floats += (size_t)L.info.wino_planes * L.info.cout_pad * cin;
            if (L.info.wino_planes == 24) {
                // small problems (a few images) have too few 2 x 4 tiles to fill the chip: those forwards use F(2x2,3x3)
                weight_floats = om::align_up(weight_floats, 4);
                L.info.wino_alt_off = (int64_t)weight_floats;
                weight_floats += (size_t)16 * L.info.cout_pad * cin;
            }
        }
        L.info.wsplit_off = L.info.wsplit_scale_off = -1;
        if (!stem) {
            // split-operand mode: the fused F(4,3) form of the stride-1 3x3 layers (conv_wino14.hip: 3 kernel rows x 6 transform
            // points = 18 planes), the direct weights of every other layer
            L.info.wsplit_off = (int64_t)split_words;
            split_words += (size_t)(L.info.wino_planes == 24 ? 18 : ks * ks) * L.info.cout_pad * cin;
            L.info.wsplit_scale_off = (int64_t)split_words;
            split_words = om::align_up(split_words + L.info.cout_pad, 4);
        }
        // ... and for the stride-1 3x3 layers their DIRECT weights as well (own per-channel exponents): the latency mode
        // (om_model_set_latency_cells) runs them through the implicit GEMM when the batch has too few tiles to fill the chip
        L.info.wsplit_direct_off = L.info.wsplit_direct_scale_off = -1;
        if (!stem && L.info.wino_planes == 24) {
            L.info.wsplit_direct_off = (int64_t)split_words;
            split_words += (size_t)ks * ks * L.info.cout_pad * cin;
            L.info.wsplit_direct_scale_off = (int64_t)split_words;
            split_words = om::align_up(split_words + L.info.cout_pad, 4);
        }
        L.info.w16_off = -1;
        if (!stem) {
            L.info.w16_off = (int64_t)weight_halfs;
            weight_halfs = om::align_up(weight_halfs + om::conv_f16_weight_halfs(L.info.cout_pad, ks, cin), 8);
        }
        L.in = in; L.out = out; L.in_div = in_div; L.out_mode = out_mode; L.up = up; L.stem = stem;
        if (res) { L.res = *res; L.has_res = true; }
        layers.push_back(L);
    }

    // conv1x1 / conv3x3 (+BN+leaky) into a fresh buffer; returns the output view
    om::View cbl(const std::string& name, om::View in, int cin, int cout, int ks, int div) {
        om::View o{new_buf(div, cout), 0};
        add(name, cin, cout, ks, 1, true, in, div, o);
        return o;
    }

    om::View neck(const std::string& prefix, om::View in, int cin, int cout, int div) {
        om::View v = cbl(prefix + ".0", in, cin, cout, 1, div);
        v = cbl(prefix + ".1", v, cout, cout * 2, 3, div);
        v = cbl(prefix + ".2", v, cout * 2, cout, 1, div);
        v = cbl(prefix + ".3", v, cout, cout * 2, 3, div);
        return cbl(prefix + ".4", v, cout * 2, cout, 1, div);
    }

    void build() {
        using om::View;
        const int A = num_anchors;
        // concat buffers: [route | backbone feature], [skip32 | skip16 | skip8 | skip4]
        const bool plus = variant == 0;
        const int cat16 = new_buf(16, 256 + 512);
        const int cat8 = new_buf(8, 128 + 256);
        const int cat4 = new_buf(4, plus ? 4 * 64 : 64 + 128);

        // ---- DarkNet-53
        View cur{new_buf(1, 32), 0};
        add("backbone.conv1", 3, 32, 3, 1, true, View{om::BUF_INPUT, 0}, 1, cur, nullptr, 0, 1, true);
        const int nblocks[7] = {0, 0, 1, 2, 8, 8, 4};
        int ch = 32, div = 1;
        View x4{}, x8{}, x16{}, x32{};
        for (int idx = 2; idx <= 6; ++idx) {
            const std::string stage = "backbone.conv" + std::to_string(idx);
            View down{new_buf(div * 2, ch * 2), 0};
            add(stage + ".0", ch, ch * 2, 3, 2, true, cur, div, down);
            div *= 2;
            cur = down;
            for (int j = 1; j <= nblocks[idx]; ++j) {
                const std::string blk = stage + "." + std::to_string(j) + ".conv.";
                View mid = cbl(blk + "0", cur, ch * 2, ch, 1, div);
                View dst{};
                const bool last = j == nblocks[idx];
                if (last && idx == 3 && !plus) dst = View{cat4, 64};     // OrienMaskYOLO: x4 sits behind route8
                else if (last && idx == 4) dst = View{cat8, 128};
                else if (last && idx == 5) dst = View{cat16, 256};
                else dst = View{new_buf(div, ch * 2), 0};
                add(blk + "1", ch, ch * 2, 3, 1, true, mid, div, dst, &cur);
                cur = dst;
            }
            if (idx == 3) x4 = cur;
            if (idx == 4) x8 = cur;
            if (idx == 5) x16 = cur;
            if (idx == 6) x32 = cur;
            ch *= 2;
        }
        (void)x8; (void)x16;

        // ---- necks and routes (fpnplus.py:77-79)
        View n32 = neck("neck32", x32, 1024, 512, 32);
        add("route32.0", 512, 256, 1, 1, true, n32, 32, View{cat16, 0}, nullptr, 1, 2);
        View n16 = neck("neck16", View{cat16, 0}, 768, 256, 16);
        add("route16.0", 256, 128, 1, 1, true, n16, 16, View{cat8, 0}, nullptr, 1, 2);
        View n8 = neck("neck8", View{cat8, 0}, 384, 128, 8);

        // ---- box heads (fpnplus.py:81-83)
        const int bbox_dim = A * (5 + num_classes);
        struct { const char* name; View in; int c; int div; int out; } heads[3] = {
            {"bbox_head32", n32, 512, 32, om::BUF_BBOX32},
            {"bbox_head16", n16, 256, 16, om::BUF_BBOX16},
            {"bbox_head8", n8, 128, 8, om::BUF_BBOX8}};
        for (auto& h : heads) {
            View t = cbl(std::string(h.name) + ".0", h.in, h.c, h.c * 2, 3, h.div);
            add(std::string(h.name) + ".1", h.c * 2, bbox_dim, 1, 1, false, t, h.div, View{h.out, 0});
        }

        // ---- orientation branch (fpnplus.py:85-88; orienmask_yolo.py:82-83 for the non-Plus model)
        if (plus) {
            add("skip32.0", 512, 64, 1, 1, true, n32, 32, View{cat4, 0}, nullptr, 1, 8);
            add("skip16.0", 256, 64, 1, 1, true, n16, 16, View{cat4, 64}, nullptr, 1, 4);
            add("skip8.0", 128, 64, 1, 1, true, n8, 8, View{cat4, 128}, nullptr, 1, 2);
            add("skip4", 128, 64, 1, 1, true, x4, 4, View{cat4, 192});
        } else {
            add("route8.0", 128, 64, 1, 1, true, n8, 8, View{cat4, 0}, nullptr, 1, 2);
        }
        View o = neck("neck4", View{cat4, 0}, plus ? 256 : 192, 128, 4);
        o = cbl("orien_head.0", o, 128, 256, 3, 4);
        o = cbl("orien_head.1", o, 256, 128, 1, 4);
        o = cbl("orien_head.2", o, 128, 256, 3, 4);
        o = cbl("orien_head.3", o, 256, 128, 1, 4);
        o = cbl("orien_head.4", o, 128, 256, 3, 4);
        add("orien_head.5", 256, A * 6, 1, 1, false, o, 4, View{om::BUF_ORIENS, 0}, nullptr, 2, 1);
        find_gathers();
    }

    // The reference up-samples the routes and skips (F.interpolate, nearest) and concatenates them with a same-resolution feature
    // (orienmask_yolo_fpnplus.py:78-86).  The plain form here lets the producer store its output replicated up x up into its slice
    // of the concat buffer; skip32 alone then writes 64 copies of every value (151 MB at bs=32, 544^2) which neck4.0 reads back.
    // In split-operand mode the producer stores ONE copy at its own resolution (`side`) and the consuming 1x1 layer reads the
    // slices where they are, up-sampling in its operand addresses (conv_igemm_split.hip: GATHER): same products, same order.
    void find_gathers() {
        for (size_t c = 0; c < layers.size(); ++c) {
            om::LayerDef& C = layers[c];
            if (C.in.buf < 0 || C.in.ch_off != 0 || C.info.ksize != 1 || C.info.stride != 1 || C.info.cin != bufs[C.in.buf].C ||
                C.info.cout_pad % 128 != 0 || C.out_mode != 0)
                continue;
            std::vector<om::LayerDef::Seg> segs;
            bool any_up = false, ok = true;
            for (size_t w = 0; w < c; ++w) {
                const om::LayerDef& P = layers[w];
                if (P.out.buf != C.in.buf) continue;
                if (P.out_mode == 2 || P.info.cout % 32 != 0) ok = false;
                any_up |= P.out_mode == 1;
                segs.push_back({P.out_mode == 1 ? (int)w : -1, P.out, P.info.cout, P.out_mode == 1 ? P.up : 1});
            }
            std::sort(segs.begin(), segs.end(), [](const om::LayerDef::Seg& a, const om::LayerDef::Seg& b) { return a.view.ch_off < b.view.ch_off; });
            int at = 0;
            for (const auto& sg : segs) { ok = ok && sg.view.ch_off == at; at += sg.channels; }
            if (!ok || !any_up || at != C.info.cin || segs.size() > 4) continue;
            // the side copy REPLACES the up-sampled slice: nobody else may read it (the same-resolution slices stay where they are)
            auto reads_upsampled = [&](const om::View& v, int channels) {
                if (v.buf != C.in.buf) return false;
                for (const auto& sg : segs)
                    if (sg.producer >= 0 && v.ch_off < sg.view.ch_off + sg.channels && sg.view.ch_off < v.ch_off + channels) return true;
                return false;
            };
            for (size_t r = 0; r < layers.size(); ++r)
                if (r != c && (reads_upsampled(layers[r].in, layers[r].info.cin) ||
                               (layers[r].has_res && reads_upsampled(layers[r].res, layers[r].info.cout))))
                    ok = false;
            if (!ok) continue;
            for (auto& sg : segs)
                if (sg.producer >= 0) layers[sg.producer].side = new_buf(layers[sg.producer].in_div, layers[sg.producer].info.cout);
            C.gather = segs;
        }
    }
    // split operands, fp32 tensors, activations not kept for om_layer_output_view (which reports a slice of the concat buffer)
    bool upsample_on_read = true;      // om_model_set_upsample_on_read
    bool gather_active(bool f16) const { return (f16 || precision == 1) && !keep_all && upsample_on_read; }

    // F(2x4,3x3) needs enough tiles to fill the chip: measured at 544^2, bs=4 is 4 % faster with F(2x2) and bs=8 is 4 % faster
    // with F(2x4); the switch is on the number of 1/32-scale cells in the batch (289 per 544^2 image).
    // With split operands (precision 1) F(2x4) runs at every size: its matrix instructions are 5.3x cheaper than the fp32-operand
    // F(2x2) kernel's, which outweighs idle workgroup slots at small batches -- and an image's results then do not depend on the
    // batch it is in.
    // Latency mode (precision 1, opt-in: 0 = off): a forward whose batch holds fewer than this many 1/32-scale cells runs its
    // stride-1 3x3 layers as direct convolutions in the implicit GEMM (small tiles, one short round) instead of the fused
    // F(4,3) kernel, whose 128 x 64 tiles leave most of the chip idle at one or two images and take cin / 16 x 6 groups of
    // ~1200 cycles each: 544^2, one image, 2.9 of the forward's 3.8 ms.  Other arithmetic than the fused kernel (same
    // tolerance against the reference), so outputs then depend on which side of the switch a batch is: off by default.
    long long latency_cells = 0;
    int latency_ksplit = 8;      // latency mode: most parts a tile's k loop is cut into (conv_igemm_split.hip; 1 = whole tiles)
    bool direct_3x3(int B, int H, int W) const {
        return precision == 1 && !keep_all && latency_cells > 0 && (long long)B * (H / 32) * (W / 32) < latency_cells;
    }
    // om_model_attach_postprocess: the step's postprocess rides on the forward.  Decode + select (which read the box heads only)
    // are launched on a SECOND stream of the library's own as soon as the last box-head layer is in the caller's stream -- forked
    // there by an event, joined behind the forward's last layer -- and the mask kernel follows on the caller's stream.  The
    // select kernel is one workgroup per image (0.11 ms at any batch size) and the decode a short pass over the heads: beside
    // the skips, neck4 and the orientation head they cost nothing.  Same kernels, same inputs: same bits as om_forward followed
    // by om_postprocess.
    struct PostAttach {
        bool on = false;
        om_post_cfg cfg;
        float* out_bbox = nullptr; int64_t* out_cls = nullptr; uint8_t* out_mask = nullptr;
        int32_t* out_count = nullptr; int32_t* out_keep = nullptr;
        void* ws = nullptr; size_t ws_bytes = 0;
    } post;
    struct Side { hipStream_t main = nullptr, side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; unsigned long long used = 0; };
    std::vector<Side> sides;      // one second stream per caller stream (batches in flight on several streams stay independent)
    Side capture_side;            // a second stream of its own for forwards issued DURING a stream capture (created by
                                  // om_model_attach_postprocess, outside any capture: nothing may be created inside one, and a caller
                                  // stream's own side stream and events may be in use by an eager step at the same time)
    unsigned long long side_clock = 0;
    // THREADING CONTRACT of one om_model (include/orienmask_hip.h says the same): forwards may be ENQUEUED from several host threads as
    // long as every thread uses its own caller stream and its own workspace; `sides` / `capture_side` / `post` are looked up and
    // changed under side_mutex, and a thread then works on a COPY of its stream's entry.  What the mutex does not cover, and the
    // caller must therefore serialise: (1) stream CAPTURES -- all captures share capture_side's one stream and event pair, so one
    // capture at a time per model; (2) profiling (om_profile_*: ev_pool / ev_used / prof_forwards are plain members) -- single-threaded,
    // which is how bench.py and the tests use it; (3) more than 64 caller streams with an attached postprocess in flight at once --
    // the 65th takes over the least recently used entry, whose side stream must have joined (it has, unless 64 forwards are still
    // being enqueued concurrently).  The Python plugin is single-threaded under the GIL and stays inside this contract.
    std::mutex side_mutex;
    int head_last = -2;      // graph index of the last bbox_head* layer (-1: none; -2: not looked up yet)
    void find_head_last() {
        head_last = -1;
        for (int l = 0; l < (int)layers.size(); ++l)
            if (std::strncmp(layers[l].info.name, "bbox_head", 9) == 0) head_last = l;
    }
    // ... per layer: only where the fused kernel would have at most 128 of its 128 x 64 tiles (half the CUs idle); with more
    // tiles it is the faster form again (136^2 128 -> 256, one image: 160 tiles, 0.057 ms against 0.080 ms direct)
    bool direct_3x3_layer(const om::LayerDef& L, int B, int H, int W) const {
        if (!direct_3x3(B, H, W) || L.info.wino_planes != 24) return false;
        int R = 0, Ct = 0, ncb = 0, nrb = 0;
        om::wino14_geometry(B, H / L.in_div, W / L.in_div, &R, &Ct, &ncb, &nrb);
        return (long long)nrb * ncb * (L.info.cout_pad / 64) <= 128;
    }

    bool use_f24(int B, int H, int W) const {
        return precision == 1 || (long long)B * (H / 32) * (W / 32) >= 1700ll;
    }

    // A/B switches of the two fusions below (process-wide; om_set_stem_fusion, or OM_NO_STEM3=1 / OM_NO_STEM2_F16=1 in the
    // environment read once): 0 = the third layer inside the split-operand stem kernel, 1 = the fp16 first-two-layers kernel
    static int& stem_fusion_flag(int which) {
        static int flags[2] = {-1, -1};
        return flags[which];
    }
    static bool stem_fusion_on(int which) {
        int& f = stem_fusion_flag(which);
        if (f < 0) {
            const char* e = std::getenv(which == 0 ? "OM_NO_STEM3" : "OM_NO_STEM2_F16");
            f = (e && e[0] == '1') ? 0 : 1;
        }
        return f != 0;
    }

    // split-operand mode: conv1 (the stem) and conv2.0 run as ONE kernel (conv_stem2.hip) when the second is the 32 -> 64 3x3
    // stride-2 layer reading the first one's output -- unless every activation is kept for om_layer_output_view
    // (never in the fp16-activation forward, whatever the precision mode says: its buffers hold 2-byte elements)
    bool stem2_fused(size_t index, bool f16 = false) const {
        if (f16 || precision != 1 || keep_all || index != 0 || layers.size() < 2 || !layers[0].stem) return false;
        const om::LayerDef& a = layers[0];
        const om::LayerDef& b = layers[1];
        return a.info.cout == 32 && b.info.cin == 32 && b.info.cout == 64 && b.info.cout_pad == 64 && b.info.ksize == 3 &&
               b.info.stride == 2 && !b.has_res && b.out_mode == 0 && b.in.buf == a.out.buf && b.in.ch_off == a.out.ch_off &&
               b.info.wsplit_off >= 0 && b.info.wino_planes != 24;
    }

    // ... and the 64 -> 32 1x1 convolution behind them (backbone.conv2.1.conv.0) inside the same kernel (round 5)
    bool stem3_fused() const {
        if (!stem_fusion_on(0) || !stem2_fused(0) || layers.size() < 3) return false;
        const om::LayerDef& b = layers[1];
        const om::LayerDef& c = layers[2];
        return c.info.cin == 64 && c.info.cout == 32 && c.info.cout_pad == 32 && c.info.ksize == 1 && c.info.stride == 1 && !c.has_res &&
               c.out_mode == 0 && c.in.buf == b.out.buf && c.in.ch_off == b.out.ch_off && c.info.wsplit_off >= 0 && c.out.buf >= 0 &&
               c.gather.empty() && c.side < 0;
    }

    // the fp16-activation forward: the same two layers as conv_stem2_f16_kernel (round 5), under the same conditions
    bool stem2_fused_f16(size_t index) const {
        if (!stem_fusion_on(1) || keep_all || index != 0 || layers.size() < 2 || !layers[0].stem) return false;
        const om::LayerDef& a = layers[0];
        const om::LayerDef& b = layers[1];
        return a.info.cout == 32 && b.info.cin == 32 && b.info.cout == 64 && b.info.cout_pad == 64 && b.info.ksize == 3 &&
               b.info.stride == 2 && !b.has_res && b.out_mode == 0 && b.in.buf == a.out.buf && b.in.ch_off == a.out.ch_off &&
               b.info.w16_off >= 0 && b.out.buf >= 0;
    }

    size_t buf_floats(int i, int B, int H, int W) const {
        return (size_t)B * (H / bufs[i].div) * (W / bufs[i].div) * bufs[i].C;
    }

    // ---- workspace layout: activations and the per-layer Winograd scratch share memory by LIVE RANGE.
    // A buffer lives from the first layer that writes it to the last layer that reads (or writes) it; a layer's transformed-input
    // scratch lives for that layer only.  Items are placed first-fit by address among the items whose ranges overlap theirs, so
    // the forward needs the peak of the live set instead of the sum of all ~95 tensors (544x544, B=32: 12.7 GiB -> see
    // DESIGN.md).  keep_all (om_model_keep_activations) gives every tensor its own slab again so that om_layer_output_view
    // can be read after the forward.
    struct Layout {
        std::vector<size_t> buf_off;        // per activation buffer
        std::vector<size_t> scratch_off;    // per layer (Winograd layers only)
        size_t tickets_off = 0, partial_off = 0, total = 0;
    };
    bool keep_all = false;

    // split-operand mode: the layers that run the two-kernel wide form of the fused 3x3 kernel (conv_wino14.hip, round 6): from 512
    // input channels on (where the pre-pass's 2.5 x the input through HBM is small next to the layer's work), whole pairs of N tiles
    // ... and only where the fused kernel's 128 x 64 tiles outnumber the CUs: while ONE round of them covers the layer, a round of half
    // as many 128 x 128 tiles takes longer (17^2 512 -> 1024: 0.14 against 0.21 ms per round; at bs = 32 the fused kernel needs 1.56
    // rounds = 0.26-0.28 ms, the wide form one round + the pre-pass = 0.22-0.23 ms)
    bool wide_3x3_layer(const om::LayerDef& L, int B, int H, int W) const {
        if (!(precision == 1 && wino14_wide_on() && L.info.wino_off >= 0 && L.info.wino_planes == 24 && L.info.ksize == 3 &&
              L.info.stride == 1 && L.out_mode == 0 && L.info.cin >= 512 && L.info.cout_pad % 128 == 0 && L.info.cout % 4 == 0) ||
            direct_3x3_layer(L, B, H, W))
            return false;
        int R = 0, Ct = 0, ncb = 0, nrb = 0;
        om::wino14_geometry(B, H / L.in_div, W / L.in_div, &R, &Ct, &ncb, &nrb);
        return (long long)nrb * ncb * (L.info.cout_pad / 64) > 256;
    }

    size_t layer_scratch_floats(const om::LayerDef& L, int B, int H, int W) const {
        if (L.info.wino_off < 0) return 0;
        if (precision == 1 && L.info.wino_planes == 24)      // conv_wino14.hip transforms its input on chip, but for the wide form's V
            return wide_3x3_layer(L, B, H, W) ? om::wino14_wide_scratch_floats(B, H / L.in_div, W / L.in_div, L.info.cin) : 0;
        return (L.info.wino_planes == 24 && use_f24(B, H, W)) ? om::wino24_scratch_floats(B, H / L.in_div, W / L.in_div, L.info.cin)
                                                               : om::wino_scratch_floats(B, H / L.in_div, W / L.in_div, L.info.cin);
    }

    Layout layout(int B, int H, int W, bool f16) const {
        const size_t esz = f16 ? 2 : 4;
        const int nb = (int)bufs.size(), nl = (int)layers.size();
        struct Item { size_t bytes; int first, last; size_t off; };
        std::vector<Item> items(nb + nl);
        for (int i = 0; i < nb; ++i) items[i] = {om::align_up(buf_floats(i, B, H, W) * esz, 256), nl, -1, 0};      // never touched: not placed
        for (int l = 0; l < nl; ++l) {
            const om::LayerDef& L = layers[l];
            auto touch = [&](int buf) {
                if (buf < 0) return;
                if (l < items[buf].first) items[buf].first = l;
                if (l > items[buf].last) items[buf].last = l;
            };
            touch(L.in.buf); touch(L.out.buf);
            if (L.has_res) touch(L.res.buf);
            if (gather_active(f16)) {
                if (L.side >= 0) touch(L.side);
                for (const auto& sg : L.gather)
                    if (sg.producer >= 0) touch(layers[sg.producer].side);
            }
            items[nb + l] = {f16 ? 0 : om::align_up(layer_scratch_floats(L, B, H, W) * sizeof(float), 256), l, l, 0};
        }
        if (keep_all)
            for (int i = 0; i < nb; ++i) { items[i].first = 0; items[i].last = nl; }
        // place in order of first use; candidates are the gaps between the already placed items that are live at the same time
        std::vector<int> order;
        for (int i = 0; i < nb + nl; ++i)
            if (items[i].bytes && items[i].last >= items[i].first) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a].first < items[b].first; });
        std::vector<int> placed;
        size_t peak = 0;
        for (int id : order) {
            Item& it = items[id];
            std::vector<std::pair<size_t, size_t>> busy;      // [off, end) of the items whose live range overlaps
            for (int o : placed)
                if (items[o].last >= it.first && items[o].first <= it.last) busy.push_back({items[o].off, items[o].off + items[o].bytes});
            std::sort(busy.begin(), busy.end());
            size_t at = 0;
            for (auto& b : busy) {
                if (at + it.bytes <= b.first) break;
                if (b.second > at) at = b.second;
            }
            it.off = at;
            if (at + it.bytes > peak) peak = at + it.bytes;
            placed.push_back(id);
        }
        Layout out;
        out.buf_off.resize(nb);
        out.scratch_off.resize(nl);
        for (int i = 0; i < nb; ++i) out.buf_off[i] = items[i].off;
        for (int l = 0; l < nl; ++l) out.scratch_off[l] = items[nb + l].off;
        out.tickets_off = peak;                 // queue word + stream-K flags per layer, then the status word (zeroed by every forward)
        out.partial_off = peak + om::align_up(((size_t)nl * om::SYNC_WORDS + om::STATUS_WORDS) * sizeof(int), 256);
        out.total = out.partial_off + (f16 ? 0 : om::SK_PARTIAL_BYTES);
        return out;
    }
};

extern "C" {

int om_version(void) { return OM_VERSION; }
const char* om_last_error(void) { return om::g_err; }

int om_model_create(om_model** out, int num_anchors, int num_classes) {
    return om_model_create_variant(out, 0, num_anchors, num_classes);
}

int om_model_create_variant(om_model** out, int variant, int num_anchors, int num_classes) {
    OM_REQUIRE(out, OM_EINVAL, "om_model_create: out is null");
    OM_REQUIRE(variant == 0 || variant == 1, OM_EINVAL, "om_model_create_variant: variant %d (0 = FPNPlus, 1 = OrienMaskYOLO)", variant);
    OM_REQUIRE(num_anchors >= 1 && num_anchors <= 3 && num_classes >= 1 &&
                   num_anchors * (5 + num_classes) <= om::HEAD_PIX_STRIDE,
               OM_EINVAL, "om_model_create: unsupported head (anchors=%d classes=%d)", num_anchors, num_classes);
    om_model* m = new om_model();
    m->num_anchors = num_anchors;
    m->num_classes = num_classes;
    m->variant = variant;
    m->build();
    *out = m;
    return OM_OK;
}

void om_model_destroy(om_model* m) {
    if (!m) return;
    for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
    if (m->capture_side.side) m->sides.push_back(m->capture_side);
    for (om_model::Side& sd : m->sides) {
        (void)hipEventDestroy(sd.ev_fork);
        (void)hipEventDestroy(sd.ev_join);
        (void)hipStreamDestroy(sd.side);
    }
    delete m;
}

int om_model_num_layers(const om_model* m) { return m ? (int)m->layers.size() : OM_EINVAL; }

int om_model_layer_info(const om_model* m, int index, om_layer_info* info) {
    OM_REQUIRE(m && info, OM_EINVAL, "om_model_layer_info: null argument");
    OM_REQUIRE(index >= 0 && index < (int)m->layers.size(), OM_EINVAL, "om_model_layer_info: index %d", index);
    *info = m->layers[index].info;
    return OM_OK;
}

size_t om_model_weight_floats(const om_model* m) { return m ? m->weight_floats : 0; }

int om_model_load_weights(om_model* m, const void* packed_dev, size_t bytes, int dtype) {
    OM_REQUIRE(m && packed_dev, OM_EINVAL, "om_model_load_weights: null argument");
    OM_REQUIRE(dtype == 0, OM_EINVAL, "om_model_load_weights: dtype %d not supported (0 = float32)", dtype);
    OM_REQUIRE(bytes == m->weight_floats * sizeof(float), OM_EINVAL,
               "om_model_load_weights: blob is %zu bytes, the graph needs %zu", bytes, m->weight_floats * sizeof(float));
    OM_REQUIRE((reinterpret_cast<uintptr_t>(packed_dev) & 15) == 0, OM_EINVAL, "om_model_load_weights: blob not 16-byte aligned");
    m->weights = static_cast<const float*>(packed_dev);
    return OM_OK;
}

size_t om_model_weight_split_words(const om_model* m) { return m ? m->split_words : 0; }

int om_model_load_weights_split(om_model* m, const void* packed_split_dev, size_t bytes) {
    OM_REQUIRE(m && packed_split_dev, OM_EINVAL, "om_model_load_weights_split: null argument");
    OM_REQUIRE(bytes == m->split_words * 4, OM_EINVAL, "om_model_load_weights_split: blob is %zu bytes, the graph needs %zu",
               bytes, m->split_words * 4);
    OM_REQUIRE((reinterpret_cast<uintptr_t>(packed_split_dev) & 15) == 0, OM_EINVAL,
               "om_model_load_weights_split: blob not 16-byte aligned");
    m->weights_split = static_cast<const float*>(packed_split_dev);
    return OM_OK;
}

int om_model_set_precision(om_model* m, int mode) {
    OM_REQUIRE(m, OM_EINVAL, "om_model_set_precision: null model");
    OM_REQUIRE(mode == 0 || mode == 1, OM_EINVAL, "om_model_set_precision: mode %d (0 = fp32 operands, 1 = split operands)", mode);
    m->precision = mode;
    return OM_OK;
}

int om_model_get_precision(const om_model* m) { return m ? m->precision : OM_EINVAL; }

int om_model_set_upsample_on_read(om_model* m, int enable) {
    OM_REQUIRE(m, OM_EINVAL, "om_model_set_upsample_on_read: null model");
    m->upsample_on_read = enable != 0;
    return OM_OK;
}

static size_t forward_workspace_bytes(const om_model* m, int B, int H, int W, bool f16) {
    if (!m || B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32) return 0;
    return m->layout(B, H, W, f16).total;
}

size_t om_forward_workspace_bytes(const om_model* m, int B, int H, int W) { return forward_workspace_bytes(m, B, H, W, false); }

size_t om_forward_status_offset(const om_model* m, int B, int H, int W) {
    if (!m || B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32) return 0;
    // om_forward's workspace only: the fp16 forward (another layout) has neither split operands nor a stream-K form
    return m->layout(B, H, W, false).tickets_off + m->layers.size() * om::SYNC_WORDS * sizeof(int);
}
size_t om_forward_f16_workspace_bytes(const om_model* m, int B, int H, int W) { return forward_workspace_bytes(m, B, H, W, true); }

static int forward_impl(om_model* m, const float* x, int B, int H, int W, float* bbox32, float* bbox16, float* bbox8,
                        float* oriens, void* workspace, size_t ws_bytes, om_stream stream_, bool f16) {
    OM_REQUIRE(m && x && bbox32 && bbox16 && bbox8 && oriens && workspace, OM_EINVAL, "om_forward: null argument");
    OM_REQUIRE(m->weights, OM_ESTATE, "om_forward: call om_model_load_weights first");
    OM_REQUIRE(!f16 || m->weights16, OM_ESTATE, "om_forward_f16: call om_model_load_weights_f16 first");
    OM_REQUIRE(f16 || m->precision == 0 || m->weights_split, OM_ESTATE,
               "om_forward: precision mode 1 needs om_model_load_weights_split first");
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, OM_EINVAL,
               "om_forward: B=%d H=%d W=%d (H and W must be positive multiples of 32)", B, H, W);
    const size_t need = forward_workspace_bytes(m, B, H, W, f16);
    OM_REQUIRE(ws_bytes >= need, OM_ENOMEM, "om_forward: workspace %zu bytes < %zu needed", ws_bytes, need);
    OM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, OM_EINVAL, "om_forward: workspace not 256-byte aligned");
    hipStream_t const main_stream = static_cast<hipStream_t>(stream_);
    const size_t esz = f16 ? 2 : 4;
    // an attached postprocess (om_model_attach_postprocess): decode + select on the library's second stream behind the last box head
    om_model::Side sd;
    om_model::PostAttach post_q;      // the attachment as it was when this forward began
    bool fused_post, early;
    {
        std::lock_guard<std::mutex> lock(m->side_mutex);
        post_q = m->post;
        fused_post = m->post.on && !f16;
        early = fused_post && m->head_last >= 0;
        if (early) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(main_stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
            if (capturing) {
                // torch.cuda.graph captures on a stream of its own, after warm-ups elsewhere: the capture's side stream is the one
                // om_model_attach_postprocess created for this purpose (the capture isolates it)
                OM_REQUIRE(m->capture_side.side, OM_ESTATE, "om_forward: stream capture with an attached postprocess, but no capture stream (re-attach)");
                sd = m->capture_side;
                sd.main = main_stream;
            } else {
                for (om_model::Side& have : m->sides)
                    if (have.main == main_stream) { have.used = ++m->side_clock; sd = have; }
                if (!sd.side) {      // first forward with an attachment on this stream
                    if (m->sides.size() >= 64) {
                        // the least recently used entry changes owner: its side stream has long joined its old caller stream
                        // (every forward ends with the join), so the stream and its events are free to serve another
                        size_t lru = 0;
                        for (size_t i = 1; i < m->sides.size(); ++i)
                            if (m->sides[i].used < m->sides[lru].used) lru = i;
                        m->sides[lru].main = main_stream;
                        m->sides[lru].used = ++m->side_clock;
                        sd = m->sides[lru];
                    } else {
                        sd.main = main_stream;
                        sd.used = ++m->side_clock;
                        OM_CHECK_HIP(hipStreamCreateWithFlags(&sd.side, hipStreamNonBlocking));
                        OM_CHECK_HIP(hipEventCreateWithFlags(&sd.ev_fork, hipEventDisableTiming));
                        OM_CHECK_HIP(hipEventCreateWithFlags(&sd.ev_join, hipEventDisableTiming));
                        m->sides.push_back(sd);
                    }
                }
            }
        }
    }
    struct JoinGuard {      // whatever path leaves the function: the caller's stream waits for the side stream's work
        om_model::Side sd; bool forked;
        void join() {
            if (!forked) return;
            forked = false;
            (void)hipEventRecord(sd.ev_join, sd.side);
            (void)hipStreamWaitEvent(sd.main, sd.ev_join, 0);
        }
        ~JoinGuard() { join(); }
    } join_guard{sd, false};

    const om_model::Layout lay = m->layout(B, H, W, f16);
    std::vector<char*> base(m->bufs.size());
    for (size_t i = 0; i < m->bufs.size(); ++i) base[i] = static_cast<char*>(workspace) + lay.buf_off[i];
    int* tickets = reinterpret_cast<int*>(static_cast<char*>(workspace) + lay.tickets_off);
    float* sk_partial = f16 ? nullptr : reinterpret_cast<float*>(static_cast<char*>(workspace) + lay.partial_off);
    int* status = tickets + m->layers.size() * om::SYNC_WORDS;      // om_forward_status_offset
    if (int rc = om::launch_zero_words(tickets, m->layers.size() * om::SYNC_WORDS + om::STATUS_WORDS, main_stream)) return rc;
    // element pointer of a view: workspace buffers hold esz-byte elements, the four outputs are always fp32
    auto ptr_of = [&](const om::View& v) -> void* {
        switch (v.buf) {
            case om::BUF_BBOX32: return bbox32;
            case om::BUF_BBOX16: return bbox16;
            case om::BUF_BBOX8: return bbox8;
            case om::BUF_ORIENS: return oriens;
            default: return base[v.buf] + (size_t)v.ch_off * esz;
        }
    };

    int fused_into_previous = 0;
    for (const om::LayerDef& L : m->layers) {
        const om_layer_info& li = L.info;
        const int layer_index = (int)(&L - m->layers.data());
        hipStream_t const stream = main_stream;
        hipEvent_t ev_stop = nullptr, ev_mid = nullptr;
        if (m->profiling && (m->prof_mask.empty() || m->prof_mask[&L - m->layers.data()])) {
            if (m->ev_used + 3 > m->ev_pool.size()) {
                for (int k = 0; k < 3; ++k) {
                    hipEvent_t e;
                    OM_CHECK_HIP(hipEventCreate(&e));
                    m->ev_pool.push_back(e);
                }
            }
            OM_CHECK_HIP(hipEventRecord(m->ev_pool[m->ev_used], stream));
            ev_mid = m->ev_pool[m->ev_used + 1];
            ev_stop = m->ev_pool[m->ev_used + 2];
            m->ev_used += 3;
        }
        struct StopGuard {
            hipEvent_t e; hipStream_t s;
            ~StopGuard() { if (e) (void)hipEventRecord(e, s); }
        } stop_guard{ev_stop, stream};
        const float* w = m->weights + li.w_off;
        const float* scale = m->weights + li.scale_off;
        const float* shift = m->weights + li.shift_off;
        const int Hin = H / L.in_div, Win = W / L.in_div;
        if (fused_into_previous > 0) {      // conv2.0 (and conv2.1.conv.0) after the fused kernel: nothing to launch (its events bracket nothing)
            --fused_into_previous;
            if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
            continue;
        }
        if (L.stem && m->stem2_fused(&L - m->layers.data(), f16)) {
            // split-operand mode: conv1 and conv2.0 as one kernel (conv_stem2.hip) -- conv1's activation never reaches memory
            const om::LayerDef& N = m->layers[(&L - m->layers.data()) + 1];
            if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
            om::Stem2Third third{};
            const bool three = m->stem3_fused();
            if (three) {
                const om::LayerDef& T = m->layers[(&L - m->layers.data()) + 2];
                third.w_split = m->weights_split + T.info.wsplit_off; third.scale_split = m->weights_split + T.info.wsplit_scale_off;
                third.shift = m->weights + T.info.shift_off; third.out = static_cast<float*>(ptr_of(T.out));
                third.cout = T.info.cout; third.leaky = T.info.leaky; third.out_pix_stride = m->pix_stride(T.out.buf);
            }
            bool three_done = three;
            int rc = om::launch_conv_stem2_split(x, B, Hin, Win, w, scale, shift, m->weights_split + N.info.wsplit_off,
                                                 m->weights_split + N.info.wsplit_scale_off, m->weights + N.info.shift_off, N.info.cout,
                                                 N.info.leaky, static_cast<float*>(ptr_of(N.out)), m->pix_stride(N.out.buf), status, stream,
                                                 three ? &third : nullptr);
            // the predicates above look at layer shapes; the launcher also has preconditions on the views (alignment, pixel
            // stride, descriptor size).  A view it refuses takes the path that was the only one before the fusion existed: first
            // without the third layer, then the separate kernels below.
            if (rc == OM_EINVAL && three) {
                three_done = false;
                rc = om::launch_conv_stem2_split(x, B, Hin, Win, w, scale, shift, m->weights_split + N.info.wsplit_off,
                                                 m->weights_split + N.info.wsplit_scale_off, m->weights + N.info.shift_off, N.info.cout,
                                                 N.info.leaky, static_cast<float*>(ptr_of(N.out)), m->pix_stride(N.out.buf), status, stream, nullptr);
            }
            if (rc == OM_OK) {
                fused_into_previous = three_done ? 2 : 1;
                continue;
            }
            if (rc != OM_EINVAL) {
                char msg[512];
                std::snprintf(msg, sizeof(msg), "%s", om::g_err);
                om::set_error("layers %s + %s: %s", li.name, N.info.name, msg);
                return rc;
            }
        } else if (L.stem && f16 && m->stem2_fused_f16(&L - m->layers.data())) {
            // fp16 activations: conv1 and conv2.0 as one kernel too (conv_stem2_f16_kernel)
            const om::LayerDef& N = m->layers[(&L - m->layers.data()) + 1];
            if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
            int rc = om::launch_conv_stem2_f16(x, B, Hin, Win, w, scale, shift, m->weights16 + N.info.w16_off,
                                               m->weights + N.info.scale_off, m->weights + N.info.shift_off, N.info.cout, N.info.leaky,
                                               ptr_of(N.out), m->pix_stride(N.out.buf), stream);
            if (rc == OM_OK) {
                fused_into_previous = 1;
                continue;
            }
            if (rc != OM_EINVAL) {      // OM_EINVAL: a view the fused launcher refuses -> the separate kernels below
                char msg[512];
                std::snprintf(msg, sizeof(msg), "%s", om::g_err);
                om::set_error("layers %s + %s: %s", li.name, N.info.name, msg);
                return rc;
            }
        }
        if (L.stem) {
            if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
            int rc = f16 ? om::launch_conv_stem_f16(x, B, Hin, Win, w, scale, shift, li.cout, ptr_of(L.out), stream)
                         : om::launch_conv_stem(x, B, Hin, Win, w, scale, shift, li.cout,
                                                static_cast<float*>(ptr_of(L.out)), stream);
            if (rc != OM_OK) return rc;
            continue;
        }
        int rc;
        if (f16) {
            om::ConvArgsH a;
            a.in = ptr_of(L.in); a.w = m->weights16 + li.w16_off; a.scale = scale; a.shift = shift;
            a.res = L.has_res ? ptr_of(L.res) : nullptr;
            a.out = ptr_of(L.out);
            a.B = B; a.H = Hin; a.W = Win; a.cin = li.cin; a.in_pix_stride = m->pix_stride(L.in.buf);
            a.Ho = Hin / li.stride; a.Wo = Win / li.stride; a.cout = li.cout; a.cout_pad = li.cout_pad;
            a.ks = li.ksize; a.stride = li.stride; a.leaky = li.leaky;
            a.res_pix_stride = L.has_res ? m->pix_stride(L.res.buf) : 0;
            a.out_pix_stride = m->pix_stride(L.out.buf);
            a.out_mode = L.out_mode; a.up = L.up;
            a.out_f32 = L.out.buf < 0 ? 1 : 0;
            a.ticket = tickets + (&L - m->layers.data()) * om::SYNC_WORDS;
            if (m->gather_active(f16)) {      // routes / skips stored once at their own resolution, read up-sampled (as in split mode below)
                if (L.side >= 0) {
                    a.out = base[L.side];
                    a.out_pix_stride = m->pix_stride(L.side); a.out_mode = 0; a.up = 1;
                }
                a.nseg = (int)L.gather.size();
                for (int g = 0; g < a.nseg; ++g) {
                    const om::LayerDef::Seg& sg = L.gather[g];
                    const int side = sg.producer >= 0 ? m->layers[sg.producer].side : -1;
                    a.seg_ptr[g] = side >= 0 ? static_cast<const void*>(base[side]) : ptr_of(sg.view);
                    a.seg_pix_stride[g] = side >= 0 ? m->pix_stride(side) : m->pix_stride(sg.view.buf);
                    a.seg_channels[g] = sg.channels; a.seg_up[g] = sg.up;
                }
            }
            if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
            rc = om::launch_conv_igemm_f16(a, stream);
        } else {
            om::ConvArgs a;
            a.in = static_cast<const float*>(ptr_of(L.in)); a.w = w; a.scale = scale; a.shift = shift;
            a.res = L.has_res ? static_cast<const float*>(ptr_of(L.res)) : nullptr;
            a.out = static_cast<float*>(ptr_of(L.out));
            a.B = B; a.H = Hin; a.W = Win; a.cin = li.cin; a.in_pix_stride = m->pix_stride(L.in.buf);
            a.Ho = Hin / li.stride; a.Wo = Win / li.stride; a.cout = li.cout; a.cout_pad = li.cout_pad;
            a.ks = li.ksize; a.stride = li.stride; a.leaky = li.leaky;
            a.res_pix_stride = L.has_res ? m->pix_stride(L.res.buf) : 0;
            a.out_pix_stride = m->pix_stride(L.out.buf);
            a.out_mode = L.out_mode; a.up = L.up;
            a.ticket = tickets + (&L - m->layers.data()) * om::SYNC_WORDS;
            a.sk_partial = sk_partial;
            a.status = status;
            a.ksplit_max = (m->precision == 1 && m->direct_3x3(B, H, W)) ? m->latency_ksplit : 0;      // latency mode only
            if (li.wino_off >= 0 && om::wino_enabled()) {
                float* wino_scratch = reinterpret_cast<float*>(static_cast<char*>(workspace) + lay.scratch_off[&L - m->layers.data()]);
                a.mid_event = ev_mid;
                if (li.wino_planes == 24 && m->precision == 1 && m->direct_3x3_layer(L, B, H, W)) {
                    // latency mode: the same layer as a direct 3x3 convolution with split operands (implicit GEMM)
                    a.w = m->weights_split + li.wsplit_direct_off;
                    a.scale = m->weights_split + li.wsplit_direct_scale_off;
                    if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
                    rc = om::launch_conv_igemm_split(a, stream);
                } else if (li.wino_planes == 24 && m->precision == 1) {
                    // split operands: the fused F(4,3) form, one kernel, no transformed input in memory
                    a.w = m->weights_split + li.wsplit_off;
                    a.scale = m->weights_split + li.wsplit_scale_off;
                    a.split = 1;
                    if (m->wide_3x3_layer(L, B, H, W) && om::wino14_wide_supported(a)) {
                        rc = om::launch_conv_wino14_wide(a, wino_scratch, stream);      // (records ev_mid between its two kernels)
                    } else {
                        if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));
                        rc = om::launch_conv_wino14_split(a, stream);
                    }
                } else if (li.wino_planes == 24 && m->use_f24(B, H, W)) {
                    a.w = m->weights + li.wino_off;
                    rc = om::launch_conv_winograd24(a, wino_scratch, stream);
                } else {
                    a.w = m->weights + (li.wino_planes == 24 ? li.wino_alt_off : li.wino_off);
                    rc = om::launch_conv_winograd(a, wino_scratch, stream);
                }
            } else {
                if (ev_mid) OM_CHECK_HIP(hipEventRecord(ev_mid, stream));     // single-kernel layer: mid == start
                if (m->precision == 1 && li.wino_planes != 24) {
                    a.w = m->weights_split + li.wsplit_off;
                    a.scale = m->weights_split + li.wsplit_scale_off;
                    if (m->gather_active(f16)) {
                        if (L.side >= 0) {          // up-sampling producer: one copy at its own resolution, read up-sampled
                            a.out = reinterpret_cast<float*>(base[L.side]);
                            a.out_pix_stride = m->pix_stride(L.side); a.out_mode = 0; a.up = 1;
                        }
                        a.nseg = (int)L.gather.size();
                        for (int g = 0; g < a.nseg; ++g) {
                            const om::LayerDef::Seg& sg = L.gather[g];
                            const int side = sg.producer >= 0 ? m->layers[sg.producer].side : -1;
                            a.seg_ptr[g] = side >= 0 ? reinterpret_cast<const float*>(base[side]) : static_cast<const float*>(ptr_of(sg.view));
                            a.seg_pix_stride[g] = side >= 0 ? m->pix_stride(side) : m->pix_stride(sg.view.buf);
                            a.seg_channels[g] = sg.channels; a.seg_up[g] = sg.up;
                        }
                    }
                    rc = om::launch_conv_igemm_split(a, stream);
                } else {
                    rc = om::launch_conv_igemm(a, stream);
                }
            }
        }
        if (rc != OM_OK) {
            char msg[512];
            std::snprintf(msg, sizeof(msg), "%s", om::g_err);
            om::set_error("layer %s: %s", li.name, msg);
            return rc;
        }
        if (early && layer_index == m->head_last) {
            // the box heads are complete in the caller's stream: decode + select beside the rest of the forward
            OM_CHECK_HIP(hipEventRecord(sd.ev_fork, main_stream));
            OM_CHECK_HIP(hipStreamWaitEvent(sd.side, sd.ev_fork, 0));
            join_guard.forked = true;
            const om_model::PostAttach& q = post_q;
            if (int prc = om_postprocess_detect(&q.cfg, bbox32, bbox16, bbox8, B, q.out_bbox, q.out_cls, q.out_count, q.out_keep, q.ws,
                                                q.ws_bytes, sd.side))
                return prc;
        }
    }
    if (m->profiling) ++m->prof_forwards;
    if (fused_post) {
        const om_model::PostAttach& q = post_q;
        join_guard.join();
        if (!early)
            if (int prc = om_postprocess_detect(&q.cfg, bbox32, bbox16, bbox8, B, q.out_bbox, q.out_cls, q.out_count, q.out_keep, q.ws,
                                                q.ws_bytes, main_stream))
                return prc;
        if (int prc = om_postprocess_assemble(&q.cfg, oriens, B, q.out_count, q.out_mask, q.ws, q.ws_bytes, main_stream)) return prc;
    }
    return OM_OK;
}

int om_forward(om_model* m, const float* x, int B, int H, int W, float* bbox32, float* bbox16, float* bbox8,
               float* oriens, void* workspace, size_t ws_bytes, om_stream stream) {
    return forward_impl(m, x, B, H, W, bbox32, bbox16, bbox8, oriens, workspace, ws_bytes, stream, false);
}

int om_forward_f16(om_model* m, const float* x, int B, int H, int W, float* bbox32, float* bbox16, float* bbox8,
                   float* oriens, void* workspace, size_t ws_bytes, om_stream stream) {
    return forward_impl(m, x, B, H, W, bbox32, bbox16, bbox8, oriens, workspace, ws_bytes, stream, true);
}

size_t om_model_weight_halfs(const om_model* m) { return m ? m->weight_halfs : 0; }

int om_model_load_weights_f16(om_model* m, const void* packed_f16_dev, size_t bytes) {
    OM_REQUIRE(m && packed_f16_dev, OM_EINVAL, "om_model_load_weights_f16: null argument");
    OM_REQUIRE(bytes == m->weight_halfs * 2, OM_EINVAL, "om_model_load_weights_f16: blob is %zu bytes, the graph needs %zu",
               bytes, m->weight_halfs * 2);
    OM_REQUIRE((reinterpret_cast<uintptr_t>(packed_f16_dev) & 15) == 0, OM_EINVAL, "om_model_load_weights_f16: blob not 16-byte aligned");
    m->weights16 = static_cast<const _Float16*>(packed_f16_dev);
    return OM_OK;
}

int om_layer_tile_f16(const om_model* m, int index, int B, int H, int W, int* bm, int* bn, int* algo) {
    OM_REQUIRE(m && bm && bn && algo, OM_EINVAL, "om_layer_tile_f16: null argument");
    OM_REQUIRE(index >= 0 && index < (int)m->layers.size(), OM_EINVAL, "om_layer_tile_f16: index %d", index);
    const om::LayerDef& L = m->layers[index];
    if (L.stem && m->stem2_fused_f16((size_t)index)) { *bm = 128; *bn = 64; *algo = 7; return OM_OK; }      // conv1 + conv2.0 in one kernel
    if (L.stem) { *bm = 0; *bn = 0; *algo = 0; return OM_OK; }
    if (index == 1 && m->stem2_fused_f16(0)) { *bm = 0; *bn = 0; *algo = 8; return OM_OK; }                // ... which this layer is part of
    const int Hin = H / L.in_div, Win = W / L.in_div;
    const int Ho = Hin / L.info.stride, Wo = Win / L.info.stride;
    om::ConvArgsH a{};
    a.B = B; a.H = Hin; a.W = Win; a.cin = L.info.cin; a.in_pix_stride = m->pix_stride(L.in.buf);
    a.cout = L.info.cout; a.cout_pad = L.info.cout_pad; a.ks = L.info.ksize; a.stride = L.info.stride; a.out_mode = L.out_mode;
    if (om::conv3x3_f16_supported(a)) {
        om::conv3x3_tile_for_f16(B * Ho * Wo, L.info.cout_pad, Wo, L.info.cin / 32, bm, bn);
        *algo = *bm == 512 ? 6 : 4;       // 6: conv3x3_f16_tall_kernel
        return OM_OK;
    }
    om::conv_tile_for_f16(B * Ho * Wo, L.info.cout_pad, L.info.cin, bm, bn);
    *algo = 1;
    if (m->gather_active(true) && !L.gather.empty()) {      // up-sampling on read: conv_igemm_f16_kernel<..., GATHER>
        if (!(*bm == 256 && *bn == 128)) *bm = 128;
        *bn = 128;
        *algo = 5;
    }
    return OM_OK;
}

int om_layer_output_view(const om_model* m, int index, int B, int H, int W, int f16, size_t* byte_offset, int* channels,
                         int* pix_stride, int* div) {
    OM_REQUIRE(m && byte_offset && channels && pix_stride && div, OM_EINVAL, "om_layer_output_view: null argument");
    OM_REQUIRE(index >= 0 && index < (int)m->layers.size(), OM_EINVAL, "om_layer_output_view: index %d", index);
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, OM_EINVAL, "om_layer_output_view: bad shape");
    const om::LayerDef& L = m->layers[index];
    OM_REQUIRE(L.out.buf >= 0, OM_EINVAL, "om_layer_output_view: layer %s writes a caller-owned head tensor", L.info.name);
    OM_REQUIRE(m->keep_all, OM_ESTATE, "om_layer_output_view: call om_model_keep_activations(m, 1) before the forward (activations "
               "share memory by live range otherwise)");
    const size_t esz = f16 ? 2 : 4;
    *byte_offset = m->layout(B, H, W, f16 != 0).buf_off[L.out.buf] + (size_t)L.out.ch_off * esz;
    *channels = L.info.cout;
    *pix_stride = m->bufs[L.out.buf].C;
    *div = m->bufs[L.out.buf].div;      // an up-sampling layer's output is stored replicated at the buffer's resolution
    return OM_OK;
}

int om_model_set_latency_cells(om_model* m, long long cells) {
    OM_REQUIRE(m && cells >= 0, OM_EINVAL, "om_model_set_latency_cells: bad argument");
    m->latency_cells = cells;
    return OM_OK;
}

int om_model_attach_postprocess(om_model* m, const om_post_cfg* cfg, float* out_bbox, int64_t* out_cls, uint8_t* out_mask,
                                int32_t* out_count, int32_t* out_keep, void* post_workspace, size_t post_ws_bytes) {
    OM_REQUIRE(m, OM_EINVAL, "om_model_attach_postprocess: null model");
    std::lock_guard<std::mutex> lock(m->side_mutex);
    if (!cfg) {
        m->post.on = false;
        return OM_OK;
    }
    OM_REQUIRE(out_bbox && out_cls && out_mask && out_count && post_workspace, OM_EINVAL, "om_model_attach_postprocess: null argument");
    OM_REQUIRE(post_ws_bytes >= om_postprocess_workspace_bytes(cfg, 1), OM_ENOMEM, "om_model_attach_postprocess: workspace too small");
    m->post.cfg = *cfg;
    m->post.out_bbox = out_bbox; m->post.out_cls = out_cls; m->post.out_mask = out_mask;
    m->post.out_count = out_count; m->post.out_keep = out_keep;
    m->post.ws = post_workspace; m->post.ws_bytes = post_ws_bytes;
    m->post.on = true;
    if (m->head_last == -2) m->find_head_last();
    if (!m->capture_side.side) {      // (attach is never called inside a capture: it is host-side set-up)
        OM_CHECK_HIP(hipStreamCreateWithFlags(&m->capture_side.side, hipStreamNonBlocking));
        OM_CHECK_HIP(hipEventCreateWithFlags(&m->capture_side.ev_fork, hipEventDisableTiming));
        OM_CHECK_HIP(hipEventCreateWithFlags(&m->capture_side.ev_join, hipEventDisableTiming));
    }
    return OM_OK;
}

int om_model_set_latency_ksplit(om_model* m, int max_parts) {
    OM_REQUIRE(m && max_parts >= 1 && max_parts <= 8, OM_EINVAL, "om_model_set_latency_ksplit: max_parts=%d (1 .. 8)", max_parts);
    m->latency_ksplit = max_parts;
    return OM_OK;
}

int om_model_keep_activations(om_model* m, int keep) {
    OM_REQUIRE(m, OM_EINVAL, "om_model_keep_activations: null model");
    m->keep_all = keep != 0;
    return OM_OK;
}

int om_layer_tile(const om_model* m, int index, int B, int H, int W, int* bm, int* bn, int* algo) {
    OM_REQUIRE(m && bm && bn && algo, OM_EINVAL, "om_layer_tile: null argument");
    OM_REQUIRE(index >= 0 && index < (int)m->layers.size(), OM_EINVAL, "om_layer_tile: index %d", index);
    const om::LayerDef& L = m->layers[index];
    if (L.stem && m->stem2_fused((size_t)index)) { *bm = 128; *bn = 64; *algo = 9; return OM_OK; }      // conv1 + conv2.0 in one kernel
    if (L.stem) { *bm = 0; *bn = 0; *algo = 0; return OM_OK; }
    if (index == 1 && m->stem2_fused(0)) { *bm = 0; *bn = 0; *algo = 10; return OM_OK; }                // ... which this layer is part of
    if (index == 2 && m->stem3_fused()) { *bm = 0; *bn = 0; *algo = 10; return OM_OK; }
    if (L.info.wino_off >= 0 && om::wino_enabled() && L.info.wino_planes == 24 && m->precision == 1 && m->direct_3x3_layer(L, B, H, W)) {
        om::conv_tile_for_split(B * (H / L.in_div) * (W / L.in_div), L.info.cout_pad, bm, bn);
        *algo = 7;
        return OM_OK;
    }
    if (L.info.wino_off >= 0 && om::wino_enabled() && L.info.wino_planes == 24 && m->precision == 1) {
        if (m->wide_3x3_layer(L, B, H, W)) { *algo = 12; *bm = 128; *bn = 128; return OM_OK; }      // two kernels: V pre-pass + 128 x 128 tile
        *algo = 8; *bm = 128; *bn = 64;
        return OM_OK;
    }
    if (L.info.wino_off >= 0 && om::wino_enabled() && L.info.wino_planes == 24 && m->use_f24(B, H, W)) {
        *algo = 5; *bm = 64; *bn = 64;
        return OM_OK;
    }
    if (L.info.wino_off >= 0 && om::wino_enabled()) {
        const int Hl = H / L.in_div, Wl = W / L.in_div;
        *algo = om::wino_fused_for(L.info.cin) ? 3 : 2;
        *bm = 64;
        *bn = *algo == 3 ? (L.info.cout_pad % 128 == 0 ? 128 : 64)
                         : om::wino_bn((long long)B * ((Hl + 1) / 2) * ((Wl + 1) / 2), L.info.cout_pad);
        return OM_OK;
    }
    const int Ho = H / L.in_div / L.info.stride, Wo = W / L.in_div / L.info.stride;
    if (m->precision == 1 && L.info.wino_planes != 24) {
        om::conv_tile_for_split(B * Ho * Wo, L.info.cout_pad, bm, bn);
        *algo = 7;
        if (m->gather_active(false) && !L.gather.empty()) { *bm = 128; *bn = 128; *algo = 11; }      // up-sampling on read
        return OM_OK;
    }
    om::conv_tile_for(B * Ho * Wo, L.info.cout_pad, bm, bn);
    *algo = 1;
    return OM_OK;
}

int om_profile_enable(om_model* m, int enable) {
    OM_REQUIRE(m, OM_EINVAL, "om_profile_enable: null model");
    m->profiling = enable != 0;
    m->prof_mask.clear();
    m->ev_used = 0;
    m->prof_forwards = 0;
    return OM_OK;
}

int om_profile_enable_layers(om_model* m, const unsigned char* layer_mask, int n_layers) {
    OM_REQUIRE(m && layer_mask, OM_EINVAL, "om_profile_enable_layers: null argument");
    OM_REQUIRE(n_layers == (int)m->layers.size(), OM_EINVAL, "om_profile_enable_layers: n_layers=%d, graph has %zu", n_layers,
               m->layers.size());
    m->profiling = true;
    m->prof_mask.assign(layer_mask, layer_mask + n_layers);
    m->ev_used = 0;
    m->prof_forwards = 0;
    return OM_OK;
}

int om_profile_read(om_model* m, float* layer_ms, float* layer_pre_ms, int n_layers, int* n_forwards) {
    OM_REQUIRE(m && layer_ms && layer_pre_ms && n_forwards, OM_EINVAL, "om_profile_read: null argument");
    OM_REQUIRE(n_layers == (int)m->layers.size(), OM_EINVAL, "om_profile_read: n_layers=%d, graph has %zu", n_layers,
               m->layers.size());
    size_t n_rec = 0;
    for (int i = 0; i < n_layers; ++i) n_rec += (m->prof_mask.empty() || m->prof_mask[i]) ? 1 : 0;
    OM_REQUIRE(m->ev_used == (size_t)m->prof_forwards * n_rec * 3, OM_ESTATE,
               "om_profile_read: a profiled forward failed part-way");
    for (int i = 0; i < n_layers; ++i) layer_ms[i] = layer_pre_ms[i] = 0.f;
    size_t e = 0;
    for (int f = 0; f < m->prof_forwards; ++f)
        for (int i = 0; i < n_layers; ++i) {
            if (!(m->prof_mask.empty() || m->prof_mask[i])) continue;
            const size_t e0 = e;
            e += 3;
            OM_CHECK_HIP(hipEventSynchronize(m->ev_pool[e0 + 2]));
            float pre = 0.f, main = 0.f;
            OM_CHECK_HIP(hipEventElapsedTime(&pre, m->ev_pool[e0], m->ev_pool[e0 + 1]));
            OM_CHECK_HIP(hipEventElapsedTime(&main, m->ev_pool[e0 + 1], m->ev_pool[e0 + 2]));
            layer_pre_ms[i] += pre;
            layer_ms[i] += main;
        }
    *n_forwards = m->prof_forwards;
    return OM_OK;
}

int om_conv2d_mode(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* w, const float* scale,
                   const float* shift, int cout, int ksize, int stride, int leaky, const float* res, int res_pix_stride,
                   float* out, int out_pix_stride, int out_mode, int up, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && stride >= 1 && H % stride == 0 && W % stride == 0, OM_EINVAL,
               "om_conv2d: bad shape");
    OM_REQUIRE(out_mode >= 0 && out_mode <= 2 && up >= 1 && (out_mode == 1 || up == 1), OM_EINVAL,
               "om_conv2d_mode: out_mode=%d up=%d (0 NHWC, 1 NHWC replicated up x up, 2 NCHW)", out_mode, up);
    om::ConvArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H / stride; a.Wo = W / stride; a.cout = cout; a.cout_pad = om::round_up(cout, 32);
    a.ks = ksize; a.stride = stride; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = out_mode; a.up = up;
    // unit-test entry only: a library-owned ticket word so that the persistent tile queue (what om_forward
    // uses, with tickets carved from the caller's workspace) is what gets tested and benchmarked
    static int* g_ticket = nullptr;
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_igemm(a, static_cast<hipStream_t>(stream));
}

int om_conv2d_split(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* w_split,
                    const float* scale_split, const float* shift, int cout, int ksize, int stride, int leaky, const float* res,
                    int res_pix_stride, float* out, int out_pix_stride, int out_mode, int up, int tile_bm, int tile_bn,
                    int32_t* status_dev, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && stride >= 1 && H % stride == 0 && W % stride == 0, OM_EINVAL,
               "om_conv2d_split: bad shape");
    OM_REQUIRE(out_mode >= 0 && out_mode <= 2 && up >= 1 && (out_mode == 1 || up == 1), OM_EINVAL,
               "om_conv2d_split: out_mode=%d up=%d (0 NHWC, 1 NHWC replicated up x up, 2 NCHW)", out_mode, up);
    om::ConvArgs a;
    a.in = in; a.w = static_cast<const float*>(w_split); a.scale = scale_split; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H / stride; a.Wo = W / stride; a.cout = cout; a.cout_pad = om::round_up(cout, 32);
    a.ks = ksize; a.stride = stride; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = out_mode; a.up = up;
    a.force_bm = tile_bm; a.force_bn = tile_bn; a.status = status_dev;
    static int* g_ticket = nullptr;      // unit-test entry only (see om_conv2d_mode)
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_igemm_split(a, static_cast<hipStream_t>(stream));
}

int om_conv2d_split_k(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* w_split,
                      const float* scale_split, const float* shift, int cout, int ksize, int stride, int leaky, const float* res,
                      int res_pix_stride, float* out, int out_pix_stride, int out_mode, int up, int tile_bm, int tile_bn,
                      int max_parts, int32_t* status_dev, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && stride >= 1 && H % stride == 0 && W % stride == 0, OM_EINVAL,
               "om_conv2d_split_k: bad shape");
    OM_REQUIRE(out_mode >= 0 && out_mode <= 2 && up >= 1 && (out_mode == 1 || up == 1), OM_EINVAL,
               "om_conv2d_split_k: out_mode=%d up=%d (0 NHWC, 1 NHWC replicated up x up, 2 NCHW)", out_mode, up);
    OM_REQUIRE(max_parts >= 1 && max_parts <= 8, OM_EINVAL, "om_conv2d_split_k: max_parts=%d (1 .. 8)", max_parts);
    om::ConvArgs a;
    a.in = in; a.w = static_cast<const float*>(w_split); a.scale = scale_split; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H / stride; a.Wo = W / stride; a.cout = cout; a.cout_pad = om::round_up(cout, 32);
    a.ks = ksize; a.stride = stride; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = out_mode; a.up = up;
    a.force_bm = tile_bm; a.force_bn = tile_bn; a.status = status_dev;
    a.ksplit_max = max_parts;
    // unit-test entry only (see om_conv2d_mode): library-owned sync words and room for 512 parts of 128 x 64 floats
    static int* g_ticket = nullptr;
    static float* g_partial = nullptr;
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (!g_partial) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_partial), (size_t)512 * 128 * 64 * sizeof(float)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    a.sk_partial = g_partial;
    return om::launch_conv_igemm_split(a, static_cast<hipStream_t>(stream));
}

int om_conv2d_split_gather(int nseg, const float* const* seg_ptr, const int* seg_channels, const int* seg_pix_stride,
                           const int* seg_up, int B, int H, int W, const void* w_split, const float* scale_split, const float* shift,
                           int cout, int leaky, float* out, int out_pix_stride, int32_t* status_dev, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && nseg >= 1 && nseg <= 4 && seg_ptr && seg_channels && seg_pix_stride && seg_up, OM_EINVAL,
               "om_conv2d_split_gather: bad shape / null segment table (nseg=%d)", nseg);
    om::ConvArgs a;
    a.in = nullptr; a.w = static_cast<const float*>(w_split); a.scale = scale_split; a.shift = shift; a.res = nullptr; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = 0; a.in_pix_stride = 0;
    a.nseg = nseg;
    for (int g = 0; g < nseg; ++g) {
        a.seg_ptr[g] = seg_ptr[g]; a.seg_channels[g] = seg_channels[g]; a.seg_pix_stride[g] = seg_pix_stride[g]; a.seg_up[g] = seg_up[g];
        a.cin += seg_channels[g];
    }
    OM_REQUIRE(a.cin > 0 && a.cin % 32 == 0, OM_EINVAL, "om_conv2d_split_gather: %d input channels", a.cin);
    a.Ho = H; a.Wo = W; a.cout = cout; a.cout_pad = om::round_up(cout, 32);
    a.ks = 1; a.stride = 1; a.leaky = leaky; a.res_pix_stride = 0;
    a.out_pix_stride = out_pix_stride; a.out_mode = 0; a.up = 1;
    a.status = status_dev;
    static int* g_ticket = nullptr;      // unit-test entry only (see om_conv2d_mode)
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_igemm_split(a, static_cast<hipStream_t>(stream));
}

int om_conv2d(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* w, const float* scale,
              const float* shift, int cout, int ksize, int stride, int leaky, const float* res, int res_pix_stride,
              float* out, int out_pix_stride, om_stream stream) {
    return om_conv2d_mode(in, B, H, W, cin, in_pix_stride, w, scale, shift, cout, ksize, stride, leaky, res, res_pix_stride, out,
                          out_pix_stride, 0, 1, stream);
}

size_t om_conv2d_winograd_scratch_bytes(int B, int H, int W, int cin) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0) return 0;
    return om::align_up(om::wino_scratch_floats(B, H, W, cin) * sizeof(float), 256);
}

int om_conv2d_winograd(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* u,
                       const float* scale, const float* shift, int cout, int leaky, const float* res,
                       int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                       om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0, OM_EINVAL, "om_conv2d_winograd: bad shape");
    OM_REQUIRE(scratch && scratch_bytes >= om_conv2d_winograd_scratch_bytes(B, H, W, cin), OM_ENOMEM,
               "om_conv2d_winograd: scratch too small");
    om::ConvArgs a;
    a.in = in; a.w = u; a.scale = scale; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H; a.Wo = W; a.cout = cout; a.cout_pad = om::round_up(cout, 64);
    a.ks = 3; a.stride = 1; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = 0; a.up = 1;
    static int* g_ticket = nullptr;
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_winograd(a, static_cast<float*>(scratch), static_cast<hipStream_t>(stream));
}

int om_conv2d_f16(const void* in, int B, int H, int W, int cin, int in_pix_stride, const void* w, const float* scale,
                  const float* shift, int cout, int ksize, int stride, int leaky, const void* res, int res_pix_stride,
                  void* out, int out_pix_stride, int out_f32, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0 && stride >= 1 && H % stride == 0 && W % stride == 0, OM_EINVAL,
               "om_conv2d_f16: bad shape");
    om::ConvArgsH a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H / stride; a.Wo = W / stride; a.cout = cout; a.cout_pad = om::round_up(cout, 32);
    a.ks = ksize; a.stride = stride; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = 0; a.up = 1; a.out_f32 = out_f32;
    static int* g_ticket = nullptr;
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_igemm_f16(a, static_cast<hipStream_t>(stream));
}

int om_conv2d_stem_f16(const float* in, int B, int H, int W, const float* w, const float* scale, const float* shift,
                       int cout, void* out, om_stream stream) {
    return om::launch_conv_stem_f16(in, B, H, W, w, scale, shift, cout, out, static_cast<hipStream_t>(stream));
}

static int conv2d_winograd24_impl(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* u,
                                  const float* scale, const float* shift, int cout, int leaky, const float* res,
                                  int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                                  om_stream stream, int split, int32_t* status_dev);

size_t om_conv2d_winograd24_scratch_bytes(int B, int H, int W, int cin) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0) return 0;
    return om::align_up(om::wino24_scratch_floats(B, H, W, cin) * sizeof(float), 256) + om::SK_PARTIAL_BYTES;
}

int om_conv2d_winograd24(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* u,
                         const float* scale, const float* shift, int cout, int leaky, const float* res,
                         int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                         om_stream stream) {
    return conv2d_winograd24_impl(in, B, H, W, cin, in_pix_stride, u, scale, shift, cout, leaky, res, res_pix_stride, out,
                                  out_pix_stride, scratch, scratch_bytes, stream, 0, nullptr);
}

static int conv2d_winograd24_impl(const float* in, int B, int H, int W, int cin, int in_pix_stride, const float* u,
                                  const float* scale, const float* shift, int cout, int leaky, const float* res,
                                  int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                                  om_stream stream, int split, int32_t* status_dev) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0, OM_EINVAL, "om_conv2d_winograd24: bad shape");
    OM_REQUIRE(scratch && scratch_bytes >= om_conv2d_winograd24_scratch_bytes(B, H, W, cin), OM_ENOMEM,
               "om_conv2d_winograd24: scratch too small");
    om::ConvArgs a;
    a.in = in; a.w = u; a.scale = scale; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H; a.Wo = W; a.cout = cout; a.cout_pad = om::round_up(cout, 64);
    a.ks = 3; a.stride = 1; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = 0; a.up = 1; a.split = split; a.status = status_dev;
    static int* g_ticket = nullptr;
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    a.sk_partial = reinterpret_cast<float*>(static_cast<char*>(scratch) + om::align_up(om::wino24_scratch_floats(B, H, W, cin) * sizeof(float), 256));
    return om::launch_conv_winograd24(a, static_cast<float*>(scratch), static_cast<hipStream_t>(stream));
}

int om_conv2d_winograd24_split(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* u_split,
                               const float* scale_split, const float* shift, int cout, int leaky, const float* res,
                               int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                               int32_t* status_dev, om_stream stream) {
    return conv2d_winograd24_impl(in, B, H, W, cin, in_pix_stride, static_cast<const float*>(u_split), scale_split, shift, cout,
                                  leaky, res, res_pix_stride, out, out_pix_stride, scratch, scratch_bytes, stream, 1, status_dev);
}

int om_conv2d_wino14_split(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* u14_split,
                           const float* scale_split, const float* shift, int cout, int leaky, const float* res,
                           int res_pix_stride, float* out, int out_pix_stride, int32_t* status_dev, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0, OM_EINVAL, "om_conv2d_wino14_split: bad shape");
    om::ConvArgs a;
    a.in = in; a.w = static_cast<const float*>(u14_split); a.scale = scale_split; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H; a.Wo = W; a.cout = cout; a.cout_pad = om::round_up(cout, 64);
    a.ks = 3; a.stride = 1; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = 0; a.up = 1; a.split = 1; a.status = status_dev;
    static int* g_ticket = nullptr;      // unit-test entry only (see om_conv2d_mode)
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_wino14_split(a, static_cast<hipStream_t>(stream));
}

size_t om_conv2d_wino14_wide_scratch_bytes(int B, int H, int W, int cin) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cin % 16) return 0;
    return om::align_up(om::wino14_wide_scratch_floats(B, H, W, cin) * sizeof(float), 256);
}

int om_conv2d_wino14_wide(const float* in, int B, int H, int W, int cin, int in_pix_stride, const void* u14_split,
                          const float* scale_split, const float* shift, int cout, int leaky, const float* res,
                          int res_pix_stride, float* out, int out_pix_stride, void* scratch, size_t scratch_bytes,
                          int32_t* status_dev, om_stream stream) {
    OM_REQUIRE(B > 0 && H > 0 && W > 0, OM_EINVAL, "om_conv2d_wino14_wide: bad shape");
    OM_REQUIRE(scratch && scratch_bytes >= om_conv2d_wino14_wide_scratch_bytes(B, H, W, cin), OM_ENOMEM,
               "om_conv2d_wino14_wide: scratch too small");
    om::ConvArgs a;
    a.in = in; a.w = static_cast<const float*>(u14_split); a.scale = scale_split; a.shift = shift; a.res = res; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.in_pix_stride = in_pix_stride;
    a.Ho = H; a.Wo = W; a.cout = cout; a.cout_pad = om::round_up(cout, 64);
    a.ks = 3; a.stride = 1; a.leaky = leaky; a.res_pix_stride = res_pix_stride;
    a.out_pix_stride = out_pix_stride; a.out_mode = 0; a.up = 1; a.split = 1; a.status = status_dev;
    static int* g_ticket = nullptr;      // unit-test entry only (see om_conv2d_mode)
    if (!g_ticket) OM_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_ticket), om::SYNC_WORDS * sizeof(int)));
    if (int rc = om::launch_zero_words(g_ticket, om::SYNC_WORDS, static_cast<hipStream_t>(stream))) return rc;
    a.ticket = g_ticket;
    return om::launch_conv_wino14_wide(a, static_cast<float*>(scratch), static_cast<hipStream_t>(stream));
}

int om_set_wino14_wide(int on) {
    OM_REQUIRE(on == 0 || on == 1, OM_EINVAL, "om_set_wino14_wide: %d", on);
    g_wino14_wide = on;
    return OM_OK;
}
int om_get_wino14_wide(void) { return wino14_wide_on() ? 1 : 0; }

int om_wino14_dual_built(void) {
#ifdef OM_WITH_W14D
    return 1;
#else
    return 0;
#endif
}

int om_set_wino14_variant(int variant) {
    OM_REQUIRE(variant == 0 || variant == 1, OM_EINVAL, "om_set_wino14_variant: %d", variant);
    OM_REQUIRE(variant == 0 || om_wino14_dual_built(), OM_EINVAL,
               "om_set_wino14_variant: this library was built without the dual-role kernel (make W14D=1)");
    om::wino14_set_variant(variant);
    return OM_OK;
}

int om_conv2d_stem3_split(const float* in, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                          const void* w2_split, const float* scale2_split, const float* shift2, int cout2, int leaky2, float* out,
                          int out_pix_stride, const void* w3_split, const float* scale3_split, const float* shift3, int cout3, int leaky3,
                          float* out3, int out3_pix_stride, int32_t* status_dev, om_stream stream) {
    om::Stem2Third third{w3_split, scale3_split, shift3, out3, cout3, leaky3, out3_pix_stride};
    return om::launch_conv_stem2_split(in, B, H, W, w1, scale1, shift1, w2_split, scale2_split, shift2, cout2, leaky2, out,
                                       out_pix_stride, status_dev, static_cast<hipStream_t>(stream), &third);
}

int om_conv2d_stem2_f16(const float* in, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                        const void* w2_f16, const float* scale2, const float* shift2, int cout2, int leaky2, void* out, int out_pix_stride,
                        om_stream stream) {
    return om::launch_conv_stem2_f16(in, B, H, W, w1, scale1, shift1, w2_f16, scale2, shift2, cout2, leaky2, out, out_pix_stride,
                                     static_cast<hipStream_t>(stream));
}

int om_get_wino14_variant(void) { return om::wino14_variant(); }

int om_set_stem_fusion(int which, int on) {
    OM_REQUIRE((which == 0 || which == 1) && (on == 0 || on == 1), OM_EINVAL, "om_set_stem_fusion: which=%d on=%d", which, on);
    om_model::stem_fusion_flag(which) = on;
    return OM_OK;
}

int om_get_stem_fusion(int which) {
    if (which != 0 && which != 1) return -1;
    return om_model::stem_fusion_on(which) ? 1 : 0;
}

int om_get_conv3x3_f16_variant(void) { return om::conv3x3_f16_get_tall(); }

int om_set_conv3x3_f16_variant(int mode) {
    OM_REQUIRE(mode >= 0 && mode <= 2, OM_EINVAL, "om_set_conv3x3_f16_variant: %d", mode);
    om::conv3x3_f16_set_tall(mode);
    return OM_OK;
}

int om_conv2d_stem(const float* in, int B, int H, int W, const float* w, const float* scale, const float* shift,
                   int cout, float* out, om_stream stream) {
    return om::launch_conv_stem(in, B, H, W, w, scale, shift, cout, out, static_cast<hipStream_t>(stream));
}

int om_conv2d_stem2_split(const float* in, int B, int H, int W, const float* w1, const float* scale1, const float* shift1,
                          const void* w2_split, const float* scale2_split, const float* shift2, int cout2, int leaky2, float* out,
                          int out_pix_stride, int32_t* status_dev, om_stream stream) {
    return om::launch_conv_stem2_split(in, B, H, W, w1, scale1, shift1, w2_split, scale2_split, shift2, cout2, leaky2, out,
                                       out_pix_stride, status_dev, static_cast<hipStream_t>(stream));
}

}  // extern "C"
