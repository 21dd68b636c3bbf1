"""CPU oracle for the OrienMask inference hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``orienmask_amd``) never does and fails loudly when its
HIP library is missing.

It restates, in torch-CPU / numpy fp32 arithmetic, what the reference computes on the
path ``OrienMaskYOLOFPNPlus.forward`` + ``OrienMaskYOLOPostProcess.__call__``.  Each
function cites the reference lines it follows.  The convolution / batch-norm /
interpolate / sigmoid / exp / sort / topk primitives live in PyTorch (a third-party
dependency of the reference, unpinned in /root/reference/requirements.txt:2); the oracle
calls the same torch CPU primitives at the same call sites and in the same memory
layouts, which is what makes it bit-identical to the reference run in this container.

Pinning: ``tools/gen_golden.py`` imports the real reference from /root/reference (this
container only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
this oracle against every one of them.  The reference itself ships no tests or golden
vectors for this path, so those fixtures are the pin.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------


_MODE = {"f16": False}     # True inside forward_f16(): fp16 activations and weights, fp32 accumulate


def _r16(t):
    """Round to the nearest IEEE fp16 value, keep float32 storage (saturating like the device store would not:
    values beyond 65504 become inf, exactly what a fp16 store does)."""
    return t.half().float()


def _cbl(sd, name, x, stride=1, res=None):
    """conv(bias=False) -> BatchNorm2d(eval, eps=1e-5) -> LeakyReLU(0.1) (+ residual).
    /root/reference/model/base.py:104-137 (ConvBNRelu), :278-279 (conv_bn_leaky); the residual add is
    /root/reference/model/backbone/darknet.py:14-15 (x + conv(x), commutative in IEEE arithmetic).

    fp16 mode (no reference counterpart; defines what om_forward_f16 computes): the input holds fp16 values, the
    weights are rounded to fp16 (not for the stem, which reads the fp32 image), the sum is fp32, BatchNorm is the
    folded fp32 scale/shift of orienmask_amd/pack.py, LeakyReLU and the residual add are fp32, one rounding to fp16."""
    w = sd[name + ".conv_block.0.weight"]
    p = name + ".conv_block.1."
    if _MODE["f16"]:
        if name != "backbone.conv1":
            w = _r16(w)
        y = F.conv2d(x, w, None, stride, w.shape[-1] // 2)
        scale = (sd[p + "weight"].double() / torch.sqrt(sd[p + "running_var"].double() + 1e-5))
        shift = (sd[p + "bias"].double() - sd[p + "running_mean"].double() * scale).float()
        y = y * scale.float().view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        y = F.leaky_relu(y, 0.1)
        return _r16(y if res is None else y + res)
    x = F.conv2d(x, w, None, stride, w.shape[-1] // 2)
    x = F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                     sd[p + "bias"], False, 0.1, 1e-5)
    x = F.leaky_relu(x, 0.1)
    return x if res is None else res + x


def _plain(sd, name, x):
    """Final 1x1 conv of a head: bias, no BN, no activation.
    /root/reference/model/orienmask_yolo_fpnplus.py:60,71.  fp16 mode: fp16 weights, fp32 sum + bias, fp32 result."""
    w = sd[name + ".weight"]
    return F.conv2d(x, _r16(w) if _MODE["f16"] else w, sd[name + ".bias"])


def _stage(sd, idx, x, nblocks):
    """One DarkNet stage: stride-2 3x3 then nblocks of x + conv3x3(conv1x1(x)).
    /root/reference/model/backbone/darknet.py:6-15,41-45."""
    p = "backbone.conv%d" % idx
    x = _cbl(sd, p + ".0", x, stride=2)
    for j in range(1, nblocks + 1):
        y = _cbl(sd, "%s.%d.conv.0" % (p, j), x)
        x = _cbl(sd, "%s.%d.conv.1" % (p, j), y, res=x)
    return x


def backbone(sd, x):
    """DarkNet53.forward, /root/reference/model/backbone/darknet.py:47-54."""
    x = _cbl(sd, "backbone.conv1", x)
    x = _stage(sd, 2, x, 1)
    x4 = _stage(sd, 3, x, 2)
    x8 = _stage(sd, 4, x4, 8)
    x16 = _stage(sd, 5, x8, 8)
    x32 = _stage(sd, 6, x16, 4)
    return x32, x16, x8, x4


def _seq(sd, prefix, x, n):
    for i in range(n):
        x = _cbl(sd, "%s.%d" % (prefix, i), x)
    return x


def _up(x, s):
    """NearestUpsample, /root/reference/model/base.py:95-101."""
    return F.interpolate(x, scale_factor=s, mode="nearest")


def forward(sd, x, num_anchors=3, return_features=False):
    """OrienMaskYOLOFPNPlus.forward, /root/reference/model/orienmask_yolo_fpnplus.py:74-90.
    Concat order: upsampled route first, backbone feature second (:78-79); skips 32,16,8,4 (:85-86)."""
    with torch.no_grad():
        x32, x16, x8, x4 = backbone(sd, x)
        n32 = _seq(sd, "neck32", x32, 5)
        n16 = _seq(sd, "neck16", torch.cat([_up(_cbl(sd, "route32.0", n32), 2), x16], 1), 5)
        n8 = _seq(sd, "neck8", torch.cat([_up(_cbl(sd, "route16.0", n16), 2), x8], 1), 5)
        b32 = _plain(sd, "bbox_head32.1", _cbl(sd, "bbox_head32.0", n32))
        b16 = _plain(sd, "bbox_head16.1", _cbl(sd, "bbox_head16.0", n16))
        b8 = _plain(sd, "bbox_head8.1", _cbl(sd, "bbox_head8.0", n8))
        cat4 = torch.cat([_up(_cbl(sd, "skip32.0", n32), 8), _up(_cbl(sd, "skip16.0", n16), 4),
                          _up(_cbl(sd, "skip8.0", n8), 2), _cbl(sd, "skip4", x4)], 1)
        o = _seq(sd, "orien_head", _seq(sd, "neck4", cat4, 5), 5)
        o = _plain(sd, "orien_head.5", o)
        o32, o16, o8 = torch.split(o, num_anchors * 2, dim=1)
    out = ((b32, o32), (b16, o16), (b8, o8))
    if return_features:
        return out, dict(x32=x32, x16=x16, x8=x8, x4=x4, neck32=n32, neck16=n16, neck8=n8, oriens=o)
    return out


def forward_f16(sd, x, num_anchors=3, return_features=False, model="OrienMaskYOLOFPNPlus"):
    """The fp16-activation configuration (BASELINE.json configs[4], SURVEY.md 8d "Config 5"): the same graph with
    every activation and convolution weight rounded to fp16 and fp32 accumulation (see _cbl).  The reference has no
    such path, so this function is a PORT-level definition, not a pinned restatement: "parity unpinned" for fp16."""
    _MODE["f16"] = True
    try:
        if model == "OrienMaskYOLO":
            return forward_yolo(sd, x, num_anchors)
        return forward(sd, x, num_anchors, return_features)
    finally:
        _MODE["f16"] = False


def forward_yolo(sd, x, num_anchors=3):
    """OrienMaskYOLO.forward (the non-Plus model), /root/reference/model/orienmask_yolo.py:71-86:
    oriens = orien_head(neck4(cat[route8(neck8), x4]))."""
    with torch.no_grad():
        x32, x16, x8, x4 = backbone(sd, x)
        n32 = _seq(sd, "neck32", x32, 5)
        n16 = _seq(sd, "neck16", torch.cat([_up(_cbl(sd, "route32.0", n32), 2), x16], 1), 5)
        n8 = _seq(sd, "neck8", torch.cat([_up(_cbl(sd, "route16.0", n16), 2), x8], 1), 5)
        b32 = _plain(sd, "bbox_head32.1", _cbl(sd, "bbox_head32.0", n32))
        b16 = _plain(sd, "bbox_head16.1", _cbl(sd, "bbox_head16.0", n16))
        b8 = _plain(sd, "bbox_head8.1", _cbl(sd, "bbox_head8.0", n8))
        cat4 = torch.cat([_up(_cbl(sd, "route8.0", n8), 2), x4], 1)
        o = _seq(sd, "orien_head", _seq(sd, "neck4", cat4, 5), 5)
        o = _plain(sd, "orien_head.5", o)
        o32, o16, o8 = torch.split(o, num_anchors * 2, dim=1)
    return (b32, o32), (b16, o16), (b8, o8)


# --------------------------------------------------------------------------------------
# NMS
# --------------------------------------------------------------------------------------

_nms_lib = None


def _load_nms_lib():
    global _nms_lib
    if _nms_lib is None:
        path = os.path.join(_HERE, "libnms_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libnms_ref.so not built; run `make -C oracle` or __graft_entry__.build()")
        lib = ctypes.CDLL(path)
        lib.nms_ref_f32.restype = ctypes.c_int
        lib.nms_ref_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _nms_lib = lib
    return _nms_lib


def nms_cpu(dets, threshold):
    """nms_cpu_kernel<float>, /root/reference/eval/src/nms_cpu.cpp:4-63, through the C
    restatement oracle/nms_ref.c.  dets [n,5] (cx,cy,w,h,score) float32; returns int64
    keep indices in ASCENDING original-index order (nms_cpu.cpp:62).  The descending
    score order comes from torch.sort, as at nms_cpu.cpp:24."""
    dets = dets.detach().to(torch.float32).contiguous()
    n = dets.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    order = torch.sort(dets[:, 4].contiguous(), 0, descending=True)[1].contiguous()
    keep_flag = np.zeros(n, dtype=np.uint8)
    lib = _load_nms_lib()
    lib.nms_ref_f32(dets.numpy().ctypes.data, order.numpy().ctypes.data, n, ctypes.c_float(threshold),
                    keep_flag.ctypes.data)
    return torch.from_numpy(np.nonzero(keep_flag)[0].astype(np.int64))


_nms_cuda_lib = None


def nms_cuda(dets, threshold):
    """nms_cuda, /root/reference/eval/src/nms_kernel.cu:72-140, through the C restatement oracle/nms_cuda_ref.c (the CUDA
    source itself cannot be built: THC headers are gone).  dets [n,5] (cx,cy,w,h,score); returns int64 keep indices in
    SCORE-DESCENDING order (order_t[keep], :136-139).  The sort (:76, torch's CUDA sort: ties unspecified) is restated as a
    stable descending sort, i.e. ties are visited in ascending index order."""
    global _nms_cuda_lib
    dets = dets.detach().to(torch.float32).contiguous()
    n = dets.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    if _nms_cuda_lib is None:
        path = os.path.join(_HERE, "libnms_cuda_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libnms_cuda_ref.so not built; run `make -C oracle` or __graft_entry__.build()")
        lib = ctypes.CDLL(path)
        lib.nms_cuda_ref_f32.restype = ctypes.c_int
        lib.nms_cuda_ref_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _nms_cuda_lib = lib
    order = torch.sort(dets[:, 4].contiguous(), stable=True, dim=0, descending=True)[1]
    sorted5 = dets[order].contiguous()
    keep_sorted = np.zeros(n, dtype=np.int64)
    m = _nms_cuda_lib.nms_cuda_ref_f32(sorted5.numpy().ctypes.data, n, ctypes.c_float(threshold), keep_sorted.ctypes.data)
    assert m >= 0
    return order[torch.from_numpy(keep_sorted[:m])]


def nms_numpy(dets, threshold, order=None):
    """Same algorithm in numpy float32 (no C); used to cross-check nms_ref.c in tests."""
    d = np.asarray(dets, dtype=np.float32)
    n = d.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    half = np.float32(2.0)
    x1 = d[:, 0] - d[:, 2] / half; y1 = d[:, 1] - d[:, 3] / half
    x2 = d[:, 0] + d[:, 2] / half; y2 = d[:, 1] + d[:, 3] / half
    areas = (x2 - x1) * (y2 - y1)
    if order is None:
        order = torch.sort(torch.from_numpy(d[:, 4].copy()), 0, descending=True)[1].numpy()
    sup = np.zeros(n, dtype=bool)
    thr = np.float32(threshold)
    for pi in range(n):
        i = order[pi]
        if sup[i]:
            continue
        js = order[pi + 1:]
        js = js[~sup[js]]
        if js.size == 0:
            continue
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[js]) - np.maximum(x1[i], x1[js]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[js]) - np.maximum(y1[i], y1[js]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[js] - inter)
        sup[js[ovr >= thr]] = True
    return np.nonzero(~sup)[0].astype(np.int64)


def batched_nms(dets, cats, threshold=0.5, normalized=True, backend="cpu"):
    """batched_nms, /root/reference/eval/function.py:77-103: boxes of class c are shifted
    by c * (1.5 + 0.5) in x and y so that different classes never overlap.  backend: which native
    kernel function.py:98-101 dispatches to ("cpu": tensors on the host, "cuda": tensors on a GPU)."""
    if dets.shape[0] == 0:
        keep = torch.zeros(0, dtype=torch.long)
    else:
        if normalized:
            max_coordinate = 1.5
        else:
            max_coordinate = dets[:, :2].max() + dets[:, 2:4].max() / 2
        shifted = dets.clone()
        shifted[:, :2] += cats.float().view(-1, 1) * (max_coordinate + 0.5)
        keep = nms_cuda(shifted, threshold) if backend == "cuda" else nms_cpu(shifted, threshold)
    return dets[keep], cats[keep], keep


# --------------------------------------------------------------------------------------
# postprocess
# --------------------------------------------------------------------------------------


class _single_thread:
    """torch's CPU sigmoid over strided rows is not bit-reproducible across thread counts (a thread's linear element range may
    start inside a row, which moves the boundary between the vectorised Sleef part and the scalar glibc tail of that row;
    tools/gen_golden.py:single_thread has the details).  The oracle evaluates the decode under one thread, as the fixtures
    were generated, so that its answer does not depend on the machine it runs on."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


class PostProcessOracle:
    """OrienMaskYOLOPostProcess, /root/reference/eval/orienmask_yolo_postprocess.py:8-166."""

    def __init__(self, grid_size, image_size, anchors, anchor_mask, num_classes, conf_thresh=0.05,
                 nms_thresh=0.5, nms_pre=400, nms_post=100, orien_thresh=0.3, nms_backend="cpu", nms_normalized=True):
        self.nms_backend = nms_backend
        self.nms_normalized = nms_normalized
        self.grids = [(int(g[0]), int(g[1])) for g in grid_size]          # (nH, nW) per scale
        if isinstance(image_size, int):
            image_size = (image_size, image_size)
        self.img_h, self.img_w = int(image_size[0]), int(image_size[1])
        self.anchor_mask = [list(m) for m in anchor_mask]
        self.num_classes = num_classes
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.nms_pre = nms_pre
        self.nms_post = nms_post
        self.orien_thresh = orien_thresh
        # constant tables, postprocess.py:17-27
        pix = torch.tensor(anchors, dtype=torch.float32)
        self.norm_anchors = torch.empty_like(pix)
        self.norm_anchors[:, 0] = pix[:, 0] / self.img_w
        self.norm_anchors[:, 1] = pix[:, 1] / self.img_h
        self.grid_anchors = self.norm_anchors.clone()
        self.grid_sizes = self.norm_anchors.clone()
        for m, (nh, nw) in zip(self.anchor_mask, self.grids):
            self.grid_anchors[m, 0] *= nw
            self.grid_anchors[m, 1] *= nh
            self.grid_sizes[m, 0] = nw
            self.grid_sizes[m, 1] = nh
        # base_xy, postprocess.py:38-45
        self.base_xy = torch.zeros(pix.shape[0], 2, self.img_h, self.img_w)
        gxs, gys, aidx = [], [], []
        for m, (nh, nw) in zip(self.anchor_mask, self.grids):
            by = torch.arange(self.img_h, dtype=torch.float) / self.img_h * nh
            bx = torch.arange(self.img_w, dtype=torch.float) / self.img_w * nw
            self.base_xy[m] = torch.stack([bx.view(1, -1).expand(self.img_h, self.img_w),
                                           by.view(-1, 1).expand(self.img_h, self.img_w)], 0)
            na = len(m)
            gy = torch.arange(nh, dtype=torch.float).view(1, nh, 1).expand(na, nh, nw).contiguous()
            gx = torch.arange(nw, dtype=torch.float).view(1, 1, nw).expand(na, nh, nw).contiguous()
            gxs.append(gx); gys.append(gy)
            aidx.append(torch.tensor(m).view(na, 1, 1).expand(na, nh, nw).contiguous())
        self.gx, self.gy = gxs, gys
        self.flat_anchor_idx = torch.cat([a.reshape(-1) for a in aidx])          # postprocess.py:59-61

    def decode_scale(self, raw, i):
        """get_boxes, postprocess.py:126-139.  raw: [A, 5+C, nH, nW] of one image."""
        nh, nw = self.grids[i]
        na = len(self.anchor_mask[i])
        t = raw.reshape(na, -1, nh, nw).permute(0, 2, 3, 1).contiguous()      # postprocess.py:86
        with _single_thread():
            obj = t[..., 4].sigmoid().view(-1)
            cls = t[..., 5:].sigmoid().view(-1, self.num_classes)
        conf = cls * obj.unsqueeze(-1)
        anc = self.norm_anchors[self.anchor_mask[i]]
        aw, ah = anc[:, 0:1], anc[:, 1:2]
        coord = t[..., 0:4]
        with _single_thread():
            sx, sy = coord[..., 0].sigmoid(), coord[..., 1].sigmoid()
        coord[..., 0] = (sx + self.gx[i]) / nw
        coord[..., 1] = (sy + self.gy[i]) / nh
        coord[..., 2] = coord[..., 2].exp() * aw.view(-1, 1, 1)
        coord[..., 3] = coord[..., 3].exp() * ah.view(-1, 1, 1)
        return coord.reshape(-1, 4), conf

    def candidates(self, predict, b):
        """Decode + threshold + top-nms_pre for image b: postprocess.py:78-114.
        Returns coord[n,4], conf[n], cls[n], anchor_idx[n], flat candidate index[n]."""
        coords, confs = [], []
        for i in range(len(self.grids)):
            c, f = self.decode_scale(predict[i][0][b], i)
            coords.append(c); confs.append(f)
        coord = torch.cat(coords, 0)
        conf = torch.cat(confs, 0)
        sel, cls = torch.nonzero(conf > self.conf_thresh, as_tuple=True)       # :102, row-major
        sel = sel.view(-1); cls = cls.view(-1)
        score = conf[sel, cls]
        if sel.numel() > self.nms_pre:                                          # :107-110
            score, top = score.topk(self.nms_pre)
            sel = sel[top]; cls = cls[top]
        return coord[sel], score, cls, self.flat_anchor_idx[sel], sel

    def orien_field(self, predict, b):
        """Upsampled orientation planes turned into pointed-to grid positions:
        postprocess.py:69-72 (bilinear x4), :92 (anchor placement), :141-144 (get_orien_grid)."""
        field = torch.zeros_like(self.base_xy)
        for i, m in enumerate(self.anchor_mask):
            up = F.interpolate(predict[i][1][b:b + 1], scale_factor=4.0, mode="bilinear", align_corners=False)[0]
            field[m] = up.view(len(m), 2, self.img_h, self.img_w)
        p = field * self.grid_anchors.view(-1, 2, 1, 1) / 2
        p += self.base_xy
        return p

    def finish(self, coord, score, cls, anchor_idx, field):
        """multi_class_nms, postprocess.py:146-166."""
        dets = torch.cat([coord, score.unsqueeze(-1)], 1)
        dets, cats, keep = batched_nms(dets, cls, self.nms_thresh, self.nms_normalized, self.nms_backend)
        if keep.numel() > self.nms_post:
            _, top = dets[:, -1].topk(self.nms_post)
            dets = dets[top]; cats = cats[top]; keep = keep[top]
        a = anchor_idx[keep]
        gsx = self.grid_sizes[a, 0]; gsy = self.grid_sizes[a, 1]
        xc = (gsx * dets[:, 0]).view(-1, 1, 1)
        yc = (gsy * dets[:, 1]).view(-1, 1, 1)
        dw = dets[:, 2].view(-1, 1, 1); dh = dets[:, 3].view(-1, 1, 1)
        masks = ((torch.abs(field[a, 0] - xc) < self.orien_thresh * dw * gsx.view(-1, 1, 1)) &
                 (torch.abs(field[a, 1] - yc) < self.orien_thresh * dh * gsy.view(-1, 1, 1)))
        return {"bbox": dets, "mask": masks, "cls": cats, "keep": keep, "anchor": a}

    def __call__(self, predict):
        """apply, postprocess.py:66-124.  predict = 3 x (bbox[B,A*(5+C),nH,nW], orien[B,2A,H/4,W/4])."""
        out = []
        with torch.no_grad():
            predict = [(p[0].detach().float().cpu(), p[1].detach().float().cpu()) for p in predict]
            for b in range(predict[0][0].shape[0]):
                coord, score, cls, aidx, sel = self.candidates(predict, b)
                res = self.finish(coord, score, cls, aidx, self.orien_field(predict, b))
                res["n_candidates"] = int(sel.numel())
                out.append(res)
        return out


def bilinear_x4_restated(plane):
    """Bilinear x4, align_corners=False, exactly as the HIP mask kernel evaluates it (numpy):
    src = (d + 0.5) * 0.25 - 0.5 clamped at 0, i0 = floor(src), i1 = min(i0 + 1, n - 1),
    row(y) = fma(v[y][x0], wx0, v[y][x1] * wx1);  out = fma(row(y0), wy0, row(y1) * wy1).
    Call site in the reference: postprocess.py:69-72 (F.interpolate).  The fma placement is
    the one torch 2.10's CPU kernel compiles to (found by search; bit-identical in this
    container), so the kernel's upsample can be compared with torch bit for bit."""
    p = np.asarray(plane, dtype=np.float32)
    h, w = p.shape

    def taps(n):
        d = np.arange(n * 4, dtype=np.float32)
        src = np.maximum((d + np.float32(0.5)) * np.float32(0.25) - np.float32(0.5), np.float32(0))
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (np.float32(1) - l1).astype(np.float32), l1

    def fma(a, b, c):     # exact product in float64, one rounding to float32
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    y0, y1, wy0, wy1 = taps(h)
    x0, x1, wx0, wx1 = taps(w)
    wx0b = np.broadcast_to(wx0[None, :], (4 * h, 4 * w)); wx1b = np.broadcast_to(wx1[None, :], (4 * h, 4 * w))
    wy0b = np.broadcast_to(wy0[:, None], (4 * h, 4 * w)); wy1b = np.broadcast_to(wy1[:, None], (4 * h, 4 * w))
    top = fma(p[y0][:, x0], wx0b, (p[y0][:, x1] * wx1b).astype(np.float32))
    bot = fma(p[y1][:, x0], wx0b, (p[y1][:, x1] * wx1b).astype(np.float32))
    return fma(top, wy0b, (bot * wy1b).astype(np.float32))


# --------------------------------------------------------------------------------------
# preprocessing
# --------------------------------------------------------------------------------------


def fast_coco_transform(image_nhwc, size=None, mean=(0, 0, 0), std=(255, 255, 255)):
    """FastCOCOTransform.__call__ with pipeline [Resize(size), Normalize(mean, std)]:
    /root/reference/data/transform.py:455-461 (permute + contiguous), :470-473 (F.interpolate bilinear,
    align_corners=False), :504-508 (sub_ mean, div_ std)."""
    x = image_nhwc.detach().float().cpu().permute(0, 3, 1, 2).contiguous()
    if size is not None:
        x = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)
    m = torch.tensor(mean, dtype=torch.float32)
    s = torch.tensor(std, dtype=torch.float32)
    x.sub_(m[:, None, None]).div_(s[:, None, None])
    return x


def pad_to_divisor(image, size_divisor=32, pad_value=0):
    """infer.pad, /root/reference/infer.py:21-32.  Returns (image, [left, right, top, down, H, W])."""
    import math
    height, width = image.shape[-2:]
    new_height = int(math.ceil(height / size_divisor) * size_divisor)
    new_width = int(math.ceil(width / size_divisor) * size_divisor)
    left, top = (new_width - width) // 2, (new_height - height) // 2
    right, down = new_width - width - left, new_height - height - top
    return F.pad(image, [left, right, top, down], value=pad_value), [left, right, top, down, new_height, new_width]


# --------------------------------------------------------------------------------------
# COCO-format conversion
# --------------------------------------------------------------------------------------


def recover_shape_bbox(bbox, sample_info):
    """COCOMetrics._recover_shape_bbox, /root/reference/eval/coco_eval.py:146-189."""
    bx, by, bw, bh = bbox[:, :4].split(1, dim=-1)
    if sample_info.get("collate_pad") is not None:
        left, right, top, down, h, w = sample_info["collate_pad"]
        nh = h - top - down
        nw = w - left - right
        bx = (bx * w - left) / nw
        by = (by * h - top) / nh
        bw = bw * w / nw
        bh = bh * h / nh
    if sample_info.get("pad") is not None:
        top, down, left, right, h, w = sample_info["pad"]
        nh = h - top - down
        nw = w - left - right
        bx = (bx * w - left) / nw
        by = (by * h - top) / nh
        bw = bw * w / nw
        bh = bh * h / nh
    if sample_info.get("hflip", False):
        bx = 1 - bx
    if sample_info.get("vflip", False):
        by = 1 - by
    oh, ow = sample_info["height"], sample_info["width"]
    return torch.cat([(bx - bw / 2) * ow, (by - bh / 2) * oh, bw * ow, bh * oh], dim=-1)


def recover_shape_segm(mask, sample_info):
    """COCOMetrics._recover_shape_segm, /root/reference/eval/coco_eval.py:191-205."""
    if sample_info.get("collate_pad") is not None:
        left, right, top, down = sample_info["collate_pad"][:4]
        mask = mask[:, top:-down if down else None, left:-right if right else None]
    if sample_info.get("pad") is not None:
        top, down, left, right = sample_info["pad"][:4]
        mask = mask[:, top:-down if down else None, left:-right if right else None]
    if sample_info.get("hflip", False):
        mask = torch.flip(mask, dims=(2,))
    if sample_info.get("vflip", False):
        mask = torch.flip(mask, dims=(1,))
    oh, ow = sample_info["height"], sample_info["width"]
    mask = F.interpolate(mask.unsqueeze(0).float(), size=(oh, ow), mode="bilinear", align_corners=False)
    return mask.squeeze(0).round().to(torch.uint8)


def rle_counts(mask2d):
    """Run lengths of pycocotools' rleEncode on a Fortran-ordered mask (call site coco_eval.py:120-122):
    column-major scan, alternating runs starting with zeros.  pycocotools 2.x (unpinned in
    /root/reference/requirements.txt:5) is not installed offline: this restates its published algorithm."""
    flat = np.asarray(mask2d, dtype=np.uint8).T.reshape(-1)          # column-major
    change = np.flatnonzero(np.diff(np.concatenate([[0], flat])))     # positions where the value flips (value before 0 is 0)
    edges = np.concatenate([[0], change, [flat.size]])
    return np.diff(edges).astype(np.int64).tolist()


_rle_lib = None


def _load_rle_lib():
    global _rle_lib
    if _rle_lib is None:
        path = os.path.join(_HERE, "librle_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/librle_ref.so not built; run `make -C oracle` or __graft_entry__.build()")
        lib = ctypes.CDLL(path)
        for f in (lib.rle_ref_encode, lib.rle_ref_to_string, lib.rle_ref_from_string):
            f.restype = ctypes.c_long
        _rle_lib = lib
    return _rle_lib


def rle_to_string(counts):
    """pycocotools rleToString through the C restatement oracle/rle_ref.c (independent of the product's Python version).
    Parity with pycocotools itself is unpinned: the library is not available offline."""
    lib = _load_rle_lib()
    c = np.ascontiguousarray(np.asarray(counts, dtype=np.int64).astype(np.uint32))
    buf = ctypes.create_string_buffer(6 * max(c.size, 1) + 1)
    n = lib.rle_ref_to_string(c.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(c.size), buf)
    return buf.raw[:n].decode("ascii")


def rle_string_decode(s, n_pixels):
    """pycocotools rleFrString (oracle/rle_ref.c): the inverse, for the round-trip property test."""
    lib = _load_rle_lib()
    out = np.zeros(len(s) + 1, dtype=np.uint32)
    m = lib.rle_ref_from_string(ctypes.c_char_p(s.encode("ascii")), out.ctypes.data_as(ctypes.c_void_p))
    counts = out[:m].astype(np.int64).tolist()
    assert sum(counts) == n_pixels
    return counts


def rle_counts_c(mask_hw):
    """pycocotools rleEncode (oracle/rle_ref.c) on one [h, w] mask: column-major runs, the first one counts zeros."""
    lib = _load_rle_lib()
    m = np.asfortranarray(np.asarray(mask_hw, dtype=np.uint8))
    flat = np.ascontiguousarray(m.reshape(-1, order="F"))
    out = np.zeros(flat.size + 1, dtype=np.uint32)
    k = lib.rle_ref_encode(flat.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(m.shape[0]), ctypes.c_long(m.shape[1]),
                           out.ctypes.data_as(ctypes.c_void_p))
    return out[:k].astype(np.int64).tolist()


