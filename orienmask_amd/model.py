"""Drop-in for the reference's ``model`` module on the inference path.

``OrienMaskYOLOFPNPlus`` keeps the reference constructor and forward() signature
(/root/reference/model/orienmask_yolo_fpnplus.py:9-10,74-90) and the reference's 524-key
state_dict, so ``build(config['model'], orienmask_amd.model)`` followed by
``load_state_dict(weights, strict=True)`` works exactly as in /root/reference/infer.py:79-83 and
/root/reference/trainer/builder.py:50-52 -- but forward() is ONE call into the HIP library
(``om_forward``): ~90 fused conv launches on torch's current stream, no torch.nn ops.

Differences a caller can observe (documented in INTEGRATION.md):
  * inference only: forward() raises in training mode and on CPU tensors (no fallback);
  * forward() returns a ``Prediction``: the reference's tuple ((bbox32, orien32), (bbox16, orien16), (bbox8, orien8)) -- it
    indexes, unpacks and iterates as one -- that also carries the forward's device-side status word, which
    ``orienmask_amd.eval.OrienMaskYOLOPostProcess`` reads together with the detection counts;
  * the three box tensors come back with the reference's SHAPE [B, A*(5+C), nH, nW] but in
    channels-last memory (NHWC, 256-float pixel stride); the orientation tensors are views of one
    contiguous [B, 6A, H/4, W/4] buffer exactly as torch.split returns them in the reference.
"""
import contextlib
import ctypes
import math

import torch
import torch.nn as nn

from . import lib as _lib
from . import pack as _pack
from .arch import model_convs, state_dict_entries

HEAD_PIX_STRIDE = 256
PRECISIONS = ("f32", "f32_split", "f16")
DEFAULT_PRECISION = "f32_split"


class Prediction(tuple):
    """What forward() returns: the reference's nested tuple of head tensors
    (/root/reference/model/orienmask_yolo_fpnplus.py:88-90) plus
      status     int32 device tensor, one word per launched sub-batch: the forward's status word(s)
                 (include/orienmask_hip.h: OM_STATUS_*), copied out of the workspace on the launch stream; None in the fp16
                 configuration, which has nothing to report
      precision  the arithmetic that produced the heads ('f32', 'f32_split', 'f16')
      rerun      callable: the same input through the fp32-operand kernels (precision 'f32') in the same workspace slot,
                 on the current stream -- what a set OM_STATUS_SPLIT_RANGE asks for; None when precision is not 'f32_split'
    Nothing here synchronises; check() does.

    Two contracts a caller of a bare forward() must know: (1) in precision 'f32_split' the range guard is
    resolved by whoever reads the status -- orienmask_amd.eval.OrienMaskYOLOPostProcess does, with its own host read; code that
    hands the heads to anything else calls check() first (an activation beyond ~6550 in front of a 3x3 layer leaves NaN heads and
    OM_STATUS_SPLIT_RANGE set; construct the model with precision='f32' to have no such condition).  (2) rerun re-reads the INPUT
    tensor of the forward: it must not be refilled in place between the forward and check() / the postprocess's collect() /
    InFlightPipeline.result() (the loops in this package allocate a fresh tensor per batch, like the reference's loader)."""

    def __new__(cls, items, status=None, precision=None, rerun=None):
        self = super().__new__(cls, items)
        self.status = status
        self.precision = precision
        self.rerun = rerun
        return self

    def flags(self):
        """Host value of the status word(s), OR-ed (one device-to-host copy; synchronises)."""
        if self.status is None:
            return 0
        out = 0
        for v in self.status.cpu().tolist():
            out |= int(v)
        return out

    def check(self):
        """Synchronous form of what the postprocess does with its own host read: returns self when the status is clean, the
        fp32-operand re-run when the split representation's range was left, raises on a stream-K time-out."""
        return resolve_status(self, self.flags())


_warned_range = False


def resolve_status(pred, flags):
    """flags: host value of pred.status.  -> the Prediction whose heads are valid."""
    global _warned_range
    if flags & _lib.OM_STATUS_SK_TIMEOUT:
        raise _lib.OrienMaskHipError("om_forward: a stream-K finisher timed out waiting for its partner workgroup "
                                     "(OM_STATUS_SK_TIMEOUT); the outputs of this batch are invalid")
    if flags & _lib.OM_STATUS_SPLIT_RANGE:
        if pred.rerun is None:
            raise _lib.OrienMaskHipError("om_forward: OM_STATUS_SPLIT_RANGE is set and the prediction has no fp32 re-run")
        if not _warned_range:
            import warnings
            warnings.warn("orienmask_amd: an activation left the fp16 range of the split-operand representation "
                          "(precision 'f32_split': |layer input| < 65504, < ~6550 in front of a stride-1 3x3 layer), or the fp32 "
                          "result itself is non-finite; this batch is re-run with fp32 operands (precision 'f32').  Set "
                          "precision='f32' for a checkpoint that does this on every batch.")
            _warned_range = True
        return pred.rerun()
    return pred


class _Node(nn.Module):
    """Pure container: holds parameters/buffers under the reference's names, never runs."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module; OrienMaskYOLOFPNPlus.forward runs the whole graph in HIP")


def _descend(root, parts):
    node = root
    for p in parts:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    return node


class OrienMaskYOLOFPNPlus(nn.Module):
    VARIANT = 0          # om_model_create_variant id

    def __init__(self, num_anchors, num_classes, pretrained=None, freeze_backbone=False,
                 backbone_batchnorm_eval=False, precision=DEFAULT_PRECISION):
        """The reference's arguments (model/orienmask_yolo_fpnplus.py:9-10) plus `precision`, which a config dict may carry
        (`build(config['model'], orienmask_amd.model)` passes every key on, trainer/builder.py:61-64): see set_precision."""
        super().__init__()
        self.num_anchors = num_anchors
        self.num_classes = num_classes
        self.pretrained = pretrained
        self.freeze_backbone = freeze_backbone
        self.backbone_batchnorm_eval = backbone_batchnorm_eval
        for spec in model_convs(type(self).__name__, num_anchors, num_classes):
            for key, shape, role in state_dict_entries(spec):
                *path, leaf = key.split(".")
                node = _descend(self, path)
                if role == "conv_w":
                    w = torch.empty(shape)
                    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                    node.register_parameter(leaf, nn.Parameter(w, requires_grad=False))
                elif role == "conv_b":
                    bound = 1.0 / math.sqrt(spec.cin * spec.ksize * spec.ksize)
                    node.register_parameter(leaf, nn.Parameter(torch.empty(shape).uniform_(-bound, bound),
                                                               requires_grad=False))
                elif role == "bn_gamma":
                    node.register_parameter(leaf, nn.Parameter(torch.ones(shape), requires_grad=False))
                elif role == "bn_beta":
                    node.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
                elif role == "bn_mean":
                    node.register_buffer(leaf, torch.zeros(shape))
                elif role == "bn_var":
                    node.register_buffer(leaf, torch.ones(shape))
                else:
                    node.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
        self._handle = None
        self._layers = None
        self._packed = None          # device blob currently bound to the handle
        self._packed16 = None        # fp16 convolution weights (precision == "f16")
        self._packed_split = None    # hi/lo fp16 pairs of the F(2x4) weights (precision == "f32_split")
        self.set_precision(precision)
        self.latency_cells = 0       # set_latency_mode(): direct 3x3 convolutions for batches of a few images (off)
        self.n_streams = 1           # set_streams(): sub-batches on side HIP streams
        self._side_streams = {}
        self._packed_device = None
        self._workspace = {}         # (device, B, H, W) -> uint8 tensor (workspace slot 0)
        self._slot = 0               # workspace_slot(): which workspace forward() uses
        self._slot_workspaces = {}   # slot > 0 -> {key: uint8 tensor}
        if pretrained is not None:
            self._load_pretrained_backbone(pretrained)

    # ------------------------------------------------------------------ weights
    def _load_pretrained_backbone(self, path):
        """BaseBackbone._load_pretrained_weights, /root/reference/model/base.py:48-64: the file is loaded INSIDE the backbone,
        so its keys are backbone-relative ('conv1.conv_block.0.weight', as in pretrained_darknet53.pth); keys that are
        missing from the backbone or have another shape are ignored and reported, as the reference does.  Returns the
        ignored keys."""
        import warnings
        sd = _pack.unwrap_checkpoint(torch.load(path, map_location="cpu", weights_only=False))
        own = self.state_dict()
        picked, ignored = {}, []
        for k, v in sd.items():
            full = "backbone." + k
            if full in own and tuple(v.shape) == tuple(own[full].shape):
                picked[full] = v
            else:
                ignored.append(k)
        if not picked:
            warnings.warn("pretrained file %s: none of its %d keys matches the backbone (expected backbone-relative keys such "
                          "as 'conv1.conv_block.0.weight'); the model keeps its initialisation" % (path, len(sd)))
        elif ignored:
            warnings.warn("pretrained file %s: ignored keys %s" % (path, ignored[:8] + (["..."] if len(ignored) > 8 else [])))
        own.update(picked)
        self.load_state_dict(own)
        return ignored

    def _ensure_handle(self):
        if self._handle is None:
            L = _lib.load()
            h = ctypes.c_void_p()
            _lib.check(L.om_model_create_variant(ctypes.byref(h), self.VARIANT, self.num_anchors, self.num_classes),
                       "om_model_create_variant")
            self._handle = h
            self._layers = _pack.graph_layers(h)
            _pack.check_graph_matches_arch(self._layers, self.num_anchors, self.num_classes, type(self).__name__)
        return self._handle

    def invalidate_packed(self):
        """Call after mutating parameters in place; load_state_dict/.to() do it themselves."""
        self._packed = None
        self._packed16 = None
        self._packed_split = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate_packed()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_packed()
        return out

    def packed_weights(self, device):
        """Device blob in the library's layout (built lazily, cached until weights change)."""
        h = self._ensure_handle()
        if self._packed is None or self._packed_device != device:
            L = _lib.load()
            total = L.om_model_weight_floats(h)
            blob = _pack.pack_state_dict(self.state_dict(), self._layers, total).to(device)
            self.bind_packed(blob)
        return self._packed

    # ------------------------------------------------------------------ fp16-activation configuration
    def set_precision(self, precision):
        """'f32_split' (the default): fp32 tensors and fp32 accumulation everywhere; every convolution but the stem carries
        each fp32 operand as a hi/lo fp16 pair and multiplies on the fp16 matrix pipe (three matrix instructions per product
        group; include/orienmask_hip.h: om_model_set_precision).  As close to the float64 answer as fp32 operands
         while activations stay inside fp16's range; a forward that leaves it raises OM_STATUS_SPLIT_RANGE on
        the device and the batch is re-run with 'f32' (Prediction.rerun; the postprocess does it with its own host read).
        'f32': fp32 operands on v_mfma_f32_32x32x2_f32, no range condition, 1.5x slower.
        'f16': fp16 activations and convolution weights, fp32 accumulation, fp32 head tensors (BASELINE.json configs[4];
        include/orienmask_hip.h: om_forward_f16) -- narrower arithmetic, never a default."""
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s, got %r" % (PRECISIONS, precision))
        self.precision = precision
        return self

    def packed_weights_split(self, device):
        h = self._ensure_handle()
        if self._packed_split is None or self._packed_split.device != device:
            total = _lib.load().om_model_weight_split_words(h)
            self.bind_packed_split(_pack.pack_state_dict_split(self.state_dict(), self._layers, total).to(device))
        return self._packed_split

    def bind_packed_split(self, blob):
        h = self._ensure_handle()
        _lib.require_cuda_tensor(blob, "packed split weights", torch.float32)
        with torch.cuda.device(blob.device):
            _lib.check(_lib.load().om_model_load_weights_split(h, ctypes.c_void_p(blob.data_ptr()), blob.numel() * 4),
                       "om_model_load_weights_split")
        self._packed_split = blob

    LATENCY_CELLS = 3500          # about twelve 544 x 544 images (289 cells each): set_latency_mode(True)

    def set_latency_mode(self, enable=True, cells=None, ksplit=None):
        """precision 'f32_split' only.  True: batches of fewer than `cells` 1/32-scale cells (default LATENCY_CELLS: up to twelve
        544 x 544 images) run their stride-1 3x3 layers as direct convolutions in the implicit GEMM instead of the fused F(4,3)
        kernel -- per layer, only where the fused kernel would have at most 128 tiles (include/orienmask_hip.h:
        om_model_set_latency_cells) -- and the implicit GEMM's launches of at most 256 tiles cut every tile's k loop into up to
        `ksplit` parts (om_model_set_latency_ksplit, default 8) run by one workgroup each, placed so that an XCD reads an eighth of
        the weights.  544^2 through the hipGraph of forward + postprocess: one image 3.2 -> 1.75 ms, two 3.4 -> 2.6, four
        4.3 -> 3.8, eight 6.4 -> 5.9; from twelve images on nothing changes.  Other summation order than the fused kernel and
        than whole tiles (same 1e-4 bar against the reference; ~1e-6 of scale apart; run-to-run identical), so outputs are no
        longer independent of the batch size: off by default, on in tester.infer_loop (the reference's bs = 1 loop)."""
        self.latency_cells = int(cells if cells is not None else self.LATENCY_CELLS) if enable else 0
        _lib.check(_lib.load().om_model_set_latency_cells(self._ensure_handle(), self.latency_cells), "om_model_set_latency_cells")
        if ksplit is not None:      # most parts a small launch's k loop is cut into (om_model_set_latency_ksplit; default 8, 1 = whole tiles)
            _lib.check(_lib.load().om_model_set_latency_ksplit(self._ensure_handle(), int(ksplit)), "om_model_set_latency_ksplit")
        return self

    def set_upsample_on_read(self, enable=True):
        """precisions 'f32_split' and 'f16': True (default) = the routes and skips store one copy at their own resolution and the 1x1
        layer behind each concat (neck16.0 / neck8.0 / neck4.0) reads them up-sampled; False = they store their output replicated
        into the concat buffer (what 'f32' does).  Bit-identical outputs; for A/B measurements and the test of that claim.
        Drops the cached workspaces (their layout differs)."""
        _lib.check(_lib.load().om_model_set_upsample_on_read(self._ensure_handle(), 1 if enable else 0), "om_model_set_upsample_on_read")
        self._workspace.clear()
        self._slot_workspaces.clear()
        return self

    def set_streams(self, n):
        """Run forward() as n independent sub-batches on n side HIP streams (batch divisible by n, else one launch).
        Tile-queue tails and per-tile prologues of one sub-batch overlap the other's kernels: +9 % forward throughput in
        the fp16 configuration at B=32, ~0 in fp32 (tools/dual_stream.py).  Per-layer profiling then times overlapping
        kernels; keep n = 1 (the default) while profiling."""
        if int(n) < 1:
            raise ValueError("n_streams must be >= 1")
        self.n_streams = int(n)
        return self

    @contextlib.contextmanager
    def workspace_slot(self, slot):
        """forward() calls inside the block use workspace number `slot` (each slot owns its activation buffers and tile
        queues), so that batches enqueued on different HIP streams can be in flight together (pipeline.InFlightPipeline).
        The model handle itself is stateless across calls: weights are only read."""
        if int(slot) < 0:
            raise ValueError("slot must be >= 0")
        prev, self._slot = self._slot, int(slot)
        try:
            yield self
        finally:
            self._slot = prev

    def packed_weights_f16(self, device):
        h = self._ensure_handle()
        if self._packed16 is None or self._packed16.device != device:
            total = _lib.load().om_model_weight_halfs(h)
            self.bind_packed_f16(_pack.pack_state_dict_f16(self.state_dict(), self._layers, total).to(device))
        return self._packed16

    def bind_packed_f16(self, blob):
        h = self._ensure_handle()
        _lib.require_cuda_tensor(blob, "packed fp16 weights", torch.float16)
        with torch.cuda.device(blob.device):
            _lib.check(_lib.load().om_model_load_weights_f16(h, ctypes.c_void_p(blob.data_ptr()), blob.numel() * 2),
                       "om_model_load_weights_f16")
        self._packed16 = blob

    def bind_packed(self, blob):
        """Bind an already packed device blob (used after an RCCL broadcast of rank 0's blob)."""
        h = self._ensure_handle()
        _lib.require_cuda_tensor(blob, "packed weights", torch.float32)
        L = _lib.load()
        with torch.cuda.device(blob.device):
            _lib.check(L.om_model_load_weights(h, ctypes.c_void_p(blob.data_ptr()), blob.numel() * 4, 0),
                       "om_model_load_weights")
        self._packed = blob
        self._packed_device = blob.device

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if self.training:
            raise RuntimeError("orienmask_amd.OrienMaskYOLOFPNPlus is inference-only: call .eval() first")
        _lib.require_cuda_tensor(x, "x", torch.float32)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError("x must be [B,3,H,W] with H and W multiples of 32, got %s" % (tuple(x.shape),))
        return self._forward(x.contiguous(), self.precision, self._slot)

    def _forward(self, x, precision, slot):
        B, _, H, W = x.shape
        dev = x.device
        L = _lib.load()
        self.packed_weights(dev)
        f16 = precision == "f16"
        if f16:
            self.packed_weights_f16(dev)
        h = self._handle
        split = precision == "f32_split"
        if split:
            self.packed_weights_split(dev)
        _lib.check(L.om_model_set_precision(h, 1 if split else 0), "om_model_set_precision")
        # sub-batches: n_streams > 1 runs B / n_streams images on each of n side streams (own workspace each) and joins
        # them on the caller's stream; images are independent, so the results are bit-identical to one launch
        n_sub = self.n_streams if (self.n_streams > 1 and B % self.n_streams == 0) else 1
        Bs = B // n_sub
        key = (dev, Bs, H, W, f16, n_sub, split)
        cache = self._workspace if slot == 0 else self._slot_workspaces.setdefault(slot, {})
        ws = cache.get(key)
        if ws is None:
            nbytes = (L.om_forward_f16_workspace_bytes if f16 else L.om_forward_workspace_bytes)(h, Bs, H, W)
            nbytes = (nbytes + 255) // 256 * 256
            for k in [k for k in cache if k[:6] != key[:6]]:      # another shape: drop; the fp32-operand re-run of a split
                del cache[k]                                      # forward keeps its workspace next to the split one
            ws = torch.empty(nbytes * n_sub, dtype=torch.uint8, device=dev)
            cache[key] = ws
        ws_each = ws.numel() // n_sub
        A = self.num_anchors
        bbox_dim = A * (5 + self.num_classes)
        heads = [torch.empty((B, H // s, W // s, HEAD_PIX_STRIDE), dtype=torch.float32, device=dev) for s in (32, 16, 8)]
        oriens = torch.empty((B, 6 * A, H // 4, W // 4), dtype=torch.float32, device=dev)
        fwd = L.om_forward_f16 if f16 else L.om_forward

        def launch(i, stream_ptr):
            xs, hs, os_ = x[i * Bs:], [t[i * Bs:] for t in heads], oriens[i * Bs:]
            return fwd(h, ctypes.c_void_p(xs.data_ptr()), Bs, H, W, ctypes.c_void_p(hs[0].data_ptr()),
                       ctypes.c_void_p(hs[1].data_ptr()), ctypes.c_void_p(hs[2].data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                       ctypes.c_void_p(ws.data_ptr() + i * ws_each), ws_each, stream_ptr)

        with torch.cuda.device(dev):
            if n_sub == 1:
                _lib.check(launch(0, _lib.current_stream_ptr(dev)), "om_forward")
            else:
                cur = torch.cuda.current_stream(dev)
                side = self._side_streams.setdefault(dev, [])
                while len(side) < n_sub:
                    side.append(torch.cuda.Stream(device=dev))
                for i in range(n_sub):
                    side[i].wait_stream(cur)
                    _lib.check(launch(i, ctypes.c_void_p(side[i].cuda_stream)), "om_forward")
                for i in range(n_sub):
                    cur.wait_stream(side[i])
            # the status word of every launched sub-batch, copied out of the workspace (which the next forward in this slot
            # clears) on the stream the caller's later work is ordered behind
            status = None
            if not f16:
                off = L.om_forward_status_offset(h, Bs, H, W)
                status = ws.view(n_sub, ws_each)[:, off:off + 4].clone().view(torch.int32).reshape(n_sub)
        bboxes = [t[..., :bbox_dim].permute(0, 3, 1, 2) for t in heads]
        o32, o16, o8 = torch.split(oriens, A * 2, dim=1)
        rerun = (lambda: self._forward(x, "f32", slot)) if split else None
        return Prediction(((bboxes[0], o32), (bboxes[1], o16), (bboxes[2], o8)), status=status, precision=precision, rerun=rerun)

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, enable=True):
        """Record one HIP event pair per layer on the launch stream during forward()."""
        _lib.check(_lib.load().om_profile_enable(self._ensure_handle(), 1 if enable else 0), "om_profile_enable")

    def keep_activations(self, keep=True):
        """Give every activation its own workspace slab (no reuse by live range) so that layer_output() can be read after a
        forward.  Drops the cached workspaces: the next forward sizes a new one."""
        _lib.check(_lib.load().om_model_keep_activations(self._ensure_handle(), 1 if keep else 0), "om_model_keep_activations")
        self._workspace.clear()
        self._slot_workspaces.clear()
        return self

    def layer_output(self, name, x_shape):
        """NCHW-shaped strided view [B, C, H/div, W/div] of layer `name`'s activation inside the workspace of the last
        forward() on an input of shape x_shape (one launch, n_streams == 1); needs keep_activations(True) before that
        forward.  For tests and debugging."""
        h = self._ensure_handle()
        idx = [l["name"] for l in self._layers].index(name)
        B, _, H, W = x_shape
        f16 = self.precision == "f16"
        ws = next(iter(self._workspace.values()))
        off, ch, ps, div = ctypes.c_size_t(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.load().om_layer_output_view(h, idx, B, H, W, 1 if f16 else 0, ctypes.byref(off), ctypes.byref(ch),
                                                    ctypes.byref(ps), ctypes.byref(div)), "om_layer_output_view")
        dt = torch.float16 if f16 else torch.float32
        esz = 2 if f16 else 4
        hh, ww = H // div.value, W // div.value
        flat = ws[off.value:off.value + ((B * hh * ww - 1) * ps.value + ch.value) * esz].view(dt)
        return torch.as_strided(flat, (B, ch.value, hh, ww), (hh * ww * ps.value, 1, ww * ps.value, ps.value))

    def profile_enable_layers(self, names):
        """Like profile_enable(True), but events are recorded only around the named layers (cheaper inside a timed region)."""
        h = self._ensure_handle()
        mask = bytes(1 if l["name"] in set(names) else 0 for l in self._layers)
        _lib.check(_lib.load().om_profile_enable_layers(h, mask, len(mask)), "om_profile_enable_layers")

    def layer_kernels(self, B, H, W):
        """Kernel instantiation that runs each layer at this size: 'conv_stem_kernel' or
        'conv_igemm_f32_kernel<BM,BN>' (graph order)."""
        h = self._ensure_handle()
        out = []
        if self.precision == "f16":
            for i, l in enumerate(self._layers):
                bm, bn, algo = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
                _lib.check(_lib.load().om_layer_tile_f16(h, i, B, H, W, ctypes.byref(bm), ctypes.byref(bn),
                                                         ctypes.byref(algo)), "om_layer_tile_f16")
                fmt = {0: "conv_stem_kernel<f16>", 1: "conv_igemm_f16_kernel<%d,%d>", 4: "conv3x3_f16_kernel<%d,%d>",
                       5: "conv_igemm_f16_kernel<%d,%d,gather>", 6: "conv3x3_f16_tall_kernel<%d,%d>",
                       7: "conv_stem2_f16_kernel<%d,%d>", 8: "(in the previous layer's kernel)"}[algo.value]
                out.append((l["name"], fmt % (bm.value, bn.value) if algo.value not in (0, 8) else fmt))
            return out
        _lib.check(_lib.load().om_model_set_precision(h, 1 if self.precision == "f32_split" else 0), "om_model_set_precision")
        for i, l in enumerate(self._layers):
            bm, bn, algo = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
            _lib.check(_lib.load().om_layer_tile(h, i, B, H, W, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(algo)),
                       "om_layer_tile")
            fmt = {0: "conv_stem_kernel", 1: "conv_igemm_f32_kernel<%d,%d>", 2: "wino_gemm_kernel<%d,%d>",
                   3: "wino_fused_kernel<%d,%d>", 5: "wino24_gemm_kernel<%d,%d>", 6: "wino24_gemm_kernel<%d,%d,split>",
                   7: "conv_igemm_split_kernel<%d,%d>", 8: "wino14_split_kernel<%d,%d>",
                   9: "conv_stem2_split_kernel<%d,%d>", 10: "(in the previous layer's kernel)",
                   11: "conv_igemm_split_kernel<%d,%d,gather>", 12: "wino14_wide_kernel<%d,%d>"}[algo.value]
            out.append((l["name"], fmt % ((bm.value, bn.value) if algo.value not in (0, 10) else ())))
        return out

    def profile_read(self):
        """(list of (layer name, main-kernel ms, pre-pass ms) summed over the recorded forwards, number of
        forwards); synchronises on the events.  The pre-pass is the Winograd input transform."""
        h = self._ensure_handle()
        n = len(self._layers)
        ms = (ctypes.c_float * n)()
        pre = (ctypes.c_float * n)()
        nf = ctypes.c_int(0)
        _lib.check(_lib.load().om_profile_read(h, ms, pre, n, ctypes.byref(nf)), "om_profile_read")
        return [(self._layers[i]["name"], float(ms[i]), float(pre[i])) for i in range(n)], nf.value

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.load().om_model_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class OrienMaskYOLO(OrienMaskYOLOFPNPlus):
    """The non-Plus model of the reference (/root/reference/model/orienmask_yolo.py:8-86):
    one up-sampling route8 concatenated with x4 into a 192-channel neck4 instead of the four skips.
    Same constructor, state_dict naming (506 keys), forward() contract and kernels; only the graph differs."""
    VARIANT = 1
