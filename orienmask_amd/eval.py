"""Drop-in for the reference's ``eval`` module on the inference path.

  OrienMaskYOLOPostProcess  same constructor and call contract as
                            /root/reference/eval/orienmask_yolo_postprocess.py:8-64
  batched_nms, nms          same signatures and return triples as
                            /root/reference/eval/function.py:55-103

Everything runs in liborienmask_hip.so (``om_postprocess`` / ``om_nms_ex``); the whole batch is three
kernel launches and ONE device->host copy (the per-image detection counts and, behind them, the forward's
status word), against the reference's per-image Python loop with four host round trips.

The reference dispatches on ``dets.is_cuda`` between two native backends with DIFFERENT semantics
(/root/reference/eval/function.py:69-72,98-101):
  "cpu"   nms_cpu.cpp:4-63     IoU >= threshold suppresses, areas (x2-x1)*(y2-y1), keep in ascending index order
  "cuda"  nms_kernel.cu:13-140 IoU >  threshold suppresses, areas w*h, keep in score-descending order
Both run here on the GPU.  The default is "cpu" -- the backend that still builds, the one the golden fixtures were
produced with; ``set_nms_backend("cuda")`` (or ``backend="cuda"`` / ``nms_backend="cuda"``) selects what a GPU user
of the reference gets.
"""
import ctypes
import functools

import torch

from . import lib as _lib
from .model import HEAD_PIX_STRIDE


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


NMS_BACKENDS = {"cpu": 0, "cuda": 1}
_nms_backend = "cpu"


def set_nms_backend(name):
    """Module default for nms / batched_nms / OrienMaskYOLOPostProcess: "cpu" or "cuda" (see the module docstring)."""
    global _nms_backend
    if name not in NMS_BACKENDS:
        raise ValueError("nms backend must be one of %s, got %r" % (sorted(NMS_BACKENDS), name))
    _nms_backend = name


def get_nms_backend():
    return _nms_backend


def nms(dets, cats, threshold=0.5, backend=None):
    """Plain NMS; returns (dets[keep], cats[keep], keep).  /root/reference/eval/function.py:55-74."""
    keep = _nms_keep(dets, threshold, backend)
    return dets[keep], cats[keep], keep


def batched_nms(dets, cats, threshold=0.5, normalized=True, backend=None):
    """Class-aware NMS: boxes are shifted by class * (max_coordinate + 0.5) so that classes never
    interact.  /root/reference/eval/function.py:77-103."""
    if dets.size(0) == 0:
        keep = dets.new_zeros(0, dtype=torch.long)
    else:
        max_coordinate = 1.5 if normalized else dets[:, :2].max() + dets[:, 2:4].max() / 2
        shifted = dets.clone()
        shifted[:, :2] += cats.float().view(-1, 1) * (max_coordinate + 0.5)
        keep = _nms_keep(shifted, threshold, backend)
    return dets[keep], cats[keep], keep


def _nms_keep(dets, threshold, backend=None):
    """om_nms_ex: the native export nms(dets[n,5], threshold) -> keep of
    /root/reference/eval/src/nms_cpu.cpp:65-75 (backend "cpu") / nms_cuda.cpp:8-17 (backend "cuda")."""
    backend = _nms_backend if backend is None else backend
    if backend not in NMS_BACKENDS:
        raise ValueError("nms backend must be one of %s, got %r" % (sorted(NMS_BACKENDS), backend))
    if dets.size(0) == 0:
        return dets.new_zeros(0, dtype=torch.long)
    _lib.require_cuda_tensor(dets, "dets")
    L = _lib.load()
    d = dets.detach().to(torch.float32).contiguous()
    n = d.shape[0]
    dev = d.device
    keep = torch.empty(n, dtype=torch.long, device=dev)
    n_keep = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = L.om_nms_workspace_bytes(n)
    if nbytes == 0:
        raise _lib.OrienMaskHipError("om_nms: %d boxes exceed the library's limit of 65536" % n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.om_nms_ex(ctypes.c_void_p(d.data_ptr()), n, float(threshold), NMS_BACKENDS[backend],
                         ctypes.c_void_p(keep.data_ptr()), ctypes.c_void_p(n_keep.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                         ws.numel(), _lib.current_stream_ptr(dev))
    _lib.check(rc, "om_nms_ex")
    return keep[:int(n_keep.item())]


class OrienMaskYOLOPostProcess:
    def __init__(self, grid_size, image_size, anchors, anchor_mask, num_classes, conf_thresh=0.05, nms_func=None,
                 nms_pre=400, nms_post=100, orien_thresh=0.3, device=None, nms_backend=None):
        """Same arguments as the reference (eval/orienmask_yolo_postprocess.py:9-11) plus `nms_backend`
        ("cpu" | "cuda" | None = the module default / the one bound into nms_func): which of the reference's two NMS
        backends the fused kernel follows."""
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else torch.device("cpu")
        self.nHs = [int(g[0]) for g in grid_size]
        self.nWs = [int(g[1]) for g in grid_size]
        self.scales = len(grid_size)
        self.image_h, self.image_w = _pair(image_size)
        self.anchors = [[float(a[0]), float(a[1])] for a in anchors]
        self.anchor_mask = [list(m) for m in anchor_mask]
        self.num_anchors = [len(m) for m in anchor_mask]
        self.num_classes = int(num_classes)
        self.conf_thresh = float(conf_thresh)
        self.nms_pre = int(nms_pre)
        self.nms_post = int(nms_post)
        self.orien_thresh = float(orien_thresh)
        self.nms = nms_func if nms_func else batched_nms
        self.nms_thresh, self.nms_normalized, bound_backend = self._nms_settings(self.nms)
        self.nms_backend = nms_backend or bound_backend or _nms_backend
        if self.nms_backend not in NMS_BACKENDS:
            raise ValueError("nms_backend must be one of %s, got %r" % (sorted(NMS_BACKENDS), self.nms_backend))
        if not 1 <= self.nms_post <= self.nms_pre <= 1024:
            raise ValueError("the fused HIP postprocess needs 1 <= nms_post <= nms_pre <= 1024 (got %d, %d)"
                             % (self.nms_post, self.nms_pre))
        # the reference takes any number of scales and anchors per scale (postprocess.py:13-36); the library's tables hold
        # 1..3 scales of 1..3 anchors each (OM_MAX_SCALES, om_post_cfg.anchors_of_scale) and 9 anchor shapes
        if not 1 <= self.scales <= _lib.OM_MAX_SCALES or len(self.anchor_mask) != self.scales:
            raise ValueError("the HIP postprocess holds 1..%d scales (got %d grids, %d anchor masks)"
                             % (_lib.OM_MAX_SCALES, self.scales, len(self.anchor_mask)))
        if any(not 1 <= n <= 3 for n in self.num_anchors):
            raise ValueError("the HIP postprocess holds 1..3 anchors per scale, got %s" % (self.num_anchors,))
        if len(self.anchors) > _lib.OM_MAX_ANCHORS or any(not 0 <= a < len(self.anchors) for m in self.anchor_mask for a in m):
            raise ValueError("at most %d anchors, anchor_mask entries index them" % _lib.OM_MAX_ANCHORS)
        self._ws = {}

    @staticmethod
    def _nms_settings(func):
        """The fused kernel implements batched_nms itself; it needs the keyword arguments that build_postprocess bound
        into the partial (/root/reference/trainer/builder.py:73-77): (threshold, normalized, backend or None)."""
        if func is batched_nms:
            return 0.5, True, None
        if isinstance(func, functools.partial) and func.func is batched_nms and not func.args:
            kw = func.keywords
            unknown = set(kw) - {"threshold", "normalized", "backend"}
            if unknown:
                raise TypeError("batched_nms got unexpected keyword arguments %s" % sorted(unknown))
            return float(kw.get("threshold", 0.5)), bool(kw.get("normalized", True)), kw.get("backend")
        # any other callable nms_func(dets, cls) -> (dets[keep], cls[keep], keep) (postprocess.py:146-148): not fusable; the
        # postprocess then runs decode + top-k on the device, calls it per image on device tensors, and builds the masks of
        # its survivors on the device (launch_foreign)
        return None, None, None

    def cfg_struct(self, bbox_pix_stride):
        c = _lib.PostCfg()
        c.num_scales = self.scales
        for i in range(self.scales):
            c.grid_h[i] = self.nHs[i]
            c.grid_w[i] = self.nWs[i]
            c.anchors_of_scale[i] = self.num_anchors[i]
            for a in range(self.num_anchors[i]):
                c.anchor_mask[i][a] = self.anchor_mask[i][a]
        c.image_h, c.image_w = self.image_h, self.image_w
        c.anchors_per_scale = self.num_anchors[0]
        for i, (w, h) in enumerate(self.anchors):
            c.anchor_w[i] = w
            c.anchor_h[i] = h
        c.num_classes = self.num_classes
        c.conf_thresh = self.conf_thresh
        c.nms_thresh = self.nms_thresh if self.nms_thresh is not None else 0.5      # unused with a foreign nms_func
        c.nms_pre, c.nms_post = self.nms_pre, self.nms_post
        c.orien_thresh = self.orien_thresh
        c.bbox_pix_stride = bbox_pix_stride
        c.nms_semantics = NMS_BACKENDS[self.nms_backend]
        c.nms_normalized = 1 if (self.nms_normalized or self.nms_normalized is None) else 0
        return c

    def __call__(self, predict):
        return self.apply(predict)

    # -- input plumbing: accept the HIP model's layouts zero-copy, anything else after one repack
    def _bbox_nhwc(self, predict):
        if len(predict) != self.scales:
            raise ValueError("predict has %d scales, the postprocess was built for %d" % (len(predict), self.scales))
        strides = set()
        for i, (bbox, _) in enumerate(predict):
            B, C, nH, nW = bbox.shape
            per = self.num_anchors[i] * (5 + self.num_classes)
            if C != per or nH != self.nHs[i] or nW != self.nWs[i]:
                raise ValueError("bbox head %d has shape %s, expected [B,%d,%d,%d]" % (i, tuple(bbox.shape), per,
                                                                                     self.nHs[i], self.nWs[i]))
            s = bbox.stride()
            if s[1] == 1 and s[3] >= C and s[2] == nW * s[3] and s[0] == nH * nW * s[3]:
                strides.add(s[3])
            else:
                strides.add(-1)
        if len(strides) == 1 and -1 not in strides:
            return [p[0] for p in predict], strides.pop()
        # one common pixel stride for every scale: the widest head's channel count
        per = max(self.num_anchors) * (5 + self.num_classes)
        out = []
        for p in predict:
            t = p[0].new_zeros((p[0].shape[0], p[0].shape[2], p[0].shape[3], per))
            t[..., :p[0].shape[1]] = p[0].permute(0, 2, 3, 1)
            out.append(t.permute(0, 3, 1, 2))
        return out, per

    def _oriens_nchw(self, predict):
        o = [p[1] for p in predict]
        oh, ow = self.image_h // 4, self.image_w // 4
        A2 = [n * 2 for n in self.num_anchors]
        tot = sum(A2)
        base = o[0]
        for i, t in enumerate(o):
            if tuple(t.shape[1:]) != (A2[i], oh, ow):
                raise ValueError("orientation head %d has shape %s, expected [B,%d,%d,%d]" % (i, tuple(t.shape), A2[i], oh, ow))
        same = all(t.stride() == (tot * oh * ow, oh * ow, ow, 1) for t in o)
        if same and all(o[i].data_ptr() == base.data_ptr() + sum(A2[:i]) * oh * ow * 4 for i in range(self.scales)):
            return base          # views of one [B, sum(A2), oh, ow] buffer (torch.split in the model): use it in place
        return torch.cat(o, dim=1).contiguous()

    def apply(self, predict):
        """Returns list[dict(bbox [K,5] f32, mask [K,H,W] bool, cls [K] i64)], K <= nms_post, per image
        (/root/reference/eval/orienmask_yolo_postprocess.py:66-124,146-166)."""
        return self.collect(self.launch(predict))

    def launch(self, predict):
        """Enqueue the three postprocess kernels on the current stream; no host synchronisation (capturable in a
        hipGraph).  Returns the raw output buffers for `collect`."""
        for p in predict:
            _lib.require_cuda_tensor(p[0], "bbox head", torch.float32)
            _lib.require_cuda_tensor(p[1], "orientation head", torch.float32)
        L = _lib.load()
        bboxes, pix_stride = self._bbox_nhwc(predict)
        oriens = self._oriens_nchw(predict)
        B = bboxes[0].shape[0]
        dev = bboxes[0].device
        cfg = self.cfg_struct(pix_stride)
        if self.nms_thresh is None:
            return self._launch_foreign(predict, bboxes, oriens, cfg, B, dev)
        while len(bboxes) < 3:
            bboxes = bboxes + [bboxes[0]]        # unused scales: never read
        key = (dev, B, pix_stride)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = L.om_postprocess_workspace_bytes(ctypes.byref(cfg), B)
            if nbytes == 0:
                _lib.check(-1, "om_postprocess_workspace_bytes")
            self._ws.clear()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        out_bbox = torch.empty((B, self.nms_post, 5), dtype=torch.float32, device=dev)
        out_cls = torch.empty((B, self.nms_post), dtype=torch.long, device=dev)
        out_mask = torch.empty((B, self.nms_post, self.image_h, self.image_w), dtype=torch.uint8, device=dev)
        # [0, B): detections per image; behind them the status word(s) of the forward that produced `predict`
        # (model.Prediction.status), so that collect()'s one device-to-host copy brings both
        status = getattr(predict, "status", None)
        n_status = 0 if status is None else int(status.numel())
        out_count = torch.empty((B + n_status,), dtype=torch.int32, device=dev)
        if n_status:
            out_count[B:].copy_(status)
        out_keep = torch.empty((B, self.nms_post), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = L.om_postprocess(ctypes.byref(cfg), ctypes.c_void_p(bboxes[0].data_ptr()),
                                  ctypes.c_void_p(bboxes[1].data_ptr()), ctypes.c_void_p(bboxes[2].data_ptr()),
                                  ctypes.c_void_p(oriens.data_ptr()), B, ctypes.c_void_p(out_bbox.data_ptr()),
                                  ctypes.c_void_p(out_cls.data_ptr()), ctypes.c_void_p(out_mask.data_ptr()),
                                  ctypes.c_void_p(out_count.data_ptr()), ctypes.c_void_p(out_keep.data_ptr()),
                                  ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.current_stream_ptr(dev))
        _lib.check(rc, "om_postprocess")
        return out_bbox, out_cls, out_mask, out_count, out_keep, (bboxes, oriens), predict     # keep the inputs alive

    def launch_step(self, model, image):
        """`self.launch(model(image))` as one sequence on the forward (tester.py:39-44's two lines): the postprocess is attached
        to the model for this forward (om_model_attach_postprocess), whose C call then launches decode + select on the library's
        second stream as soon as the box heads are written -- beside the rest of the forward -- and the mask kernel behind its last
        layer.  Same kernels, same inputs, same bits; no host synchronisation (capturable).  Falls back to the two calls where
        the fused form does not apply (a foreign nms_func, the fp16 configuration, sub-batch streams, another model class)."""
        from .model import HEAD_PIX_STRIDE, OrienMaskYOLOFPNPlus
        fused = (self.nms_thresh is not None and isinstance(model, OrienMaskYOLOFPNPlus) and model.precision != "f16"
                 and model.n_streams == 1 and self.scales == 3 and image.is_cuda and image.dim() == 4
                 and self.num_anchors == [model.num_anchors] * 3 and self.num_classes == model.num_classes
                 and [image.shape[2] // s for s in (32, 16, 8)] == self.nHs and [image.shape[3] // s for s in (32, 16, 8)] == self.nWs
                 and (self.image_h, self.image_w) == tuple(image.shape[2:]))
        if not fused:
            return self.launch(model(image))
        L = _lib.load()
        B, dev = image.shape[0], image.device
        cfg = self.cfg_struct(HEAD_PIX_STRIDE)
        ws = self._workspace(cfg, B, dev, HEAD_PIX_STRIDE)
        out_bbox = torch.empty((B, self.nms_post, 5), dtype=torch.float32, device=dev)
        out_cls = torch.empty((B, self.nms_post), dtype=torch.long, device=dev)
        out_mask = torch.empty((B, self.nms_post, self.image_h, self.image_w), dtype=torch.uint8, device=dev)
        n_status = 1
        out_count = torch.empty((B + n_status,), dtype=torch.int32, device=dev)
        out_keep = torch.empty((B, self.nms_post), dtype=torch.int32, device=dev)
        h = model._ensure_handle()
        with torch.cuda.device(dev):
            _lib.check(L.om_model_attach_postprocess(h, ctypes.byref(cfg), ctypes.c_void_p(out_bbox.data_ptr()),
                                                     ctypes.c_void_p(out_cls.data_ptr()), ctypes.c_void_p(out_mask.data_ptr()),
                                                     ctypes.c_void_p(out_count.data_ptr()), ctypes.c_void_p(out_keep.data_ptr()),
                                                     ctypes.c_void_p(ws.data_ptr()), ws.numel()), "om_model_attach_postprocess")
            try:
                predict = model(image)
            finally:
                _lib.check(L.om_model_attach_postprocess(h, None, None, None, None, None, None, None, 0), "om_model_attach_postprocess")
        status = getattr(predict, "status", None)
        if status is None or int(status.numel()) != n_status:
            raise _lib.OrienMaskHipError("launch_step: the forward returned %s status words" % (None if status is None else status.numel()))
        out_count[B:].copy_(status)
        bboxes, _ = self._bbox_nhwc(predict)
        return out_bbox, out_cls, out_mask, out_count, out_keep, (bboxes, self._oriens_nchw(predict)), predict

    def _workspace(self, cfg, B, dev, pix_stride):
        L = _lib.load()
        key = (dev, B, pix_stride)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = L.om_postprocess_workspace_bytes(ctypes.byref(cfg), B)
            if nbytes == 0:
                _lib.check(-1, "om_postprocess_workspace_bytes")
            self._ws.clear()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def _launch_foreign(self, predict, bboxes, oriens, cfg, B, dev):
        """A caller-supplied nms_func (postprocess.py:9-11,146-154): candidates on the device (om_postprocess_candidates), the
        callable per image on device tensors -- this step reads the candidate counts, so it synchronises, like the reference's
        own per-image loop -- then the survivors' masks on the device (om_postprocess_masks).  Returns the same tuple as launch."""
        L = _lib.load()
        st = _lib.current_stream_ptr(dev)
        ws = self._workspace(cfg, B, dev, cfg.bbox_pix_stride)
        heads = [ctypes.c_void_p(t.data_ptr()) for t in bboxes] + [None] * (3 - len(bboxes))
        cd = torch.empty((B, self.nms_pre, 5), dtype=torch.float32, device=dev)
        cc = torch.empty((B, self.nms_pre), dtype=torch.long, device=dev)
        cf = torch.empty((B, self.nms_pre), dtype=torch.int32, device=dev)
        cn = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.om_postprocess_candidates(ctypes.byref(cfg), heads[0], heads[1], heads[2], B, ctypes.c_void_p(cd.data_ptr()),
                                                   ctypes.c_void_p(cc.data_ptr()), ctypes.c_void_p(cf.data_ptr()),
                                                   ctypes.c_void_p(cn.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), st),
                       "om_postprocess_candidates")
        out_bbox = torch.zeros((B, self.nms_post, 5), dtype=torch.float32, device=dev)
        out_cls = torch.zeros((B, self.nms_post), dtype=torch.long, device=dev)
        out_field = torch.zeros((B, self.nms_post), dtype=torch.int32, device=dev)
        out_keep = torch.zeros((B, self.nms_post), dtype=torch.int32, device=dev)
        counts = []
        for b, n in enumerate(cn.cpu().tolist()):
            if n == 0:
                counts.append(0)
                continue
            dets, cats, keep = self.nms(cd[b, :n], cc[b, :n])
            if keep.numel() > self.nms_post:                      # postprocess.py:150-154
                _, topk = dets[:, -1].topk(self.nms_post)
                dets, cats, keep = dets[topk], cats[topk], keep[topk]
            k = int(keep.numel())
            counts.append(k)
            out_bbox[b, :k] = dets
            out_cls[b, :k] = cats
            out_field[b, :k] = cf[b, :n][keep]
            out_keep[b, :k] = keep.to(torch.int32)
        status = getattr(predict, "status", None)
        n_status = 0 if status is None else int(status.numel())
        out_count = torch.empty((B + n_status,), dtype=torch.int32, device=dev)
        out_count[:B] = torch.tensor(counts, dtype=torch.int32)
        if n_status:
            out_count[B:].copy_(status)
        out_mask = torch.empty((B, self.nms_post, self.image_h, self.image_w), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.om_postprocess_masks(ctypes.byref(cfg), ctypes.c_void_p(oriens.data_ptr()), B, ctypes.c_void_p(out_bbox.data_ptr()),
                                              ctypes.c_void_p(out_field.data_ptr()), ctypes.c_void_p(out_count.data_ptr()),
                                              ctypes.c_void_p(out_mask.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), st),
                       "om_postprocess_masks")
        return out_bbox, out_cls, out_mask, out_count, out_keep, (bboxes, oriens, out_field), predict

    def collect(self, outs):
        """The one host synchronisation of the batch: read the per-image counts (and the forward's status word behind them),
        slice the outputs.  A forward that left the range of the split-operand representation (OM_STATUS_SPLIT_RANGE,
        model.resolve_status) is repeated here with fp32 operands and postprocessed again; a stream-K time-out raises."""
        out_bbox, out_cls, out_mask, out_count, out_keep = outs[:5]
        B = out_bbox.shape[0]
        host = out_count.cpu().tolist()
        counts = host[:B]
        flags = 0
        for v in host[B:]:
            flags |= int(v)
        if flags:
            from .model import resolve_status
            return self.apply(resolve_status(outs[6], flags))
        mask_bool = out_mask.view(torch.bool)
        self.last_keep = [out_keep[b, :k] for b, k in enumerate(counts)]
        return [{"bbox": out_bbox[b, :k], "mask": mask_bool[b, :k], "cls": out_cls[b, :k]}
                for b, k in enumerate(counts)]
