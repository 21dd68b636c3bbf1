"""Drop-in for the reference's GPU preprocessing.

  FastCOCOTransform(pipeline, use_cuda)   /root/reference/data/transform.py:444-510
  pad(image, size_divisor=32, pad_value=0) /root/reference/infer.py:21-32
  build_transform(config)                  /root/reference/trainer/builder.py:108-115

The reference's pipeline objects (Resize / ShortEdgeResize / Normalize) are kept as plain
parameter holders with the same constructor arguments; ``FastCOCOTransform.__call__`` runs the
whole pipeline as ONE HIP kernel (``om_preprocess``): HWC read once, resized + normalised NCHW
written once.  ``padded(image)`` additionally fuses ``pad`` into the same kernel and returns
``(image, pad_info)`` like ``infer.pad`` does.
"""
import ctypes
import math

import torch

from . import lib as _lib


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class FastCOCOTransform:
    class Resize:
        def __init__(self, size, interpolation="bilinear", align_corners=False):
            assert isinstance(size, int) or len(size) == 2
            if interpolation != "bilinear" or align_corners:
                raise NotImplementedError("the HIP preprocess implements bilinear, align_corners=False")
            self.size = _pair(size)

        def target(self, h, w):
            return self.size

    class ShortEdgeResize:
        def __init__(self, short_length, max_size, interpolation="bilinear", align_corners=False):
            if interpolation != "bilinear" or align_corners:
                raise NotImplementedError("the HIP preprocess implements bilinear, align_corners=False")
            self.short_length = short_length
            self.max_size = max_size

        def target(self, h, w):
            scale = min(self.short_length / min(h, w), self.max_size / max(h, w))
            return int(h * scale + 0.5), int(w * scale + 0.5)

    class Normalize:
        def __init__(self, mean, std):
            self.mean = [float(m) for m in mean]
            self.std = [float(s) for s in std]

    def __init__(self, pipeline, use_cuda=True):
        if not use_cuda:
            raise _lib.OrienMaskHipError("orienmask_amd.FastCOCOTransform runs on the GPU only (no CPU fallback)")
        self.pipeline = list(pipeline)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        resizes = [t for t in self.pipeline if isinstance(t, (self.Resize, self.ShortEdgeResize))]
        norms = [t for t in self.pipeline if isinstance(t, self.Normalize)]
        if len(resizes) > 1 or len(norms) > 1 or len(resizes) + len(norms) != len(self.pipeline):
            raise NotImplementedError("pipeline must be [Resize|ShortEdgeResize]? + [Normalize]?")
        if resizes and norms and self.pipeline.index(resizes[0]) > self.pipeline.index(norms[0]):
            raise NotImplementedError("Normalize before Resize is not fused")
        self._resize = resizes[0] if resizes else None
        self._norm = norms[0] if norms else None

    def _run(self, image, divisor, pad_value):
        _lib.require_cuda_tensor(image, "image", torch.float32)
        if image.dim() != 4 or image.shape[3] != 3:
            raise ValueError("image must be [n,h,w,3] float32, got %s" % (tuple(image.shape),))
        image = image.contiguous()
        n, h, w, _ = image.shape
        rh, rw = self._resize.target(h, w) if self._resize else (h, w)
        if divisor:
            oh = int(math.ceil(rh / divisor) * divisor)
            ow = int(math.ceil(rw / divisor) * divisor)
        else:
            oh, ow = rh, rw
        left, top = (ow - rw) // 2, (oh - rh) // 2
        right, down = ow - rw - left, oh - rh - top
        mean = (ctypes.c_float * 3)(*(self._norm.mean if self._norm else [0.0] * 3))
        std = (ctypes.c_float * 3)(*(self._norm.std if self._norm else [1.0] * 3))
        out = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=image.device)
        L = _lib.load()
        with torch.cuda.device(image.device):
            rc = L.om_preprocess(ctypes.c_void_p(image.data_ptr()), n, h, w, rh, rw, mean, std, top, left, oh, ow,
                                 float(pad_value), ctypes.c_void_p(out.data_ptr()), _lib.current_stream_ptr(image.device))
        _lib.check(rc, "om_preprocess")
        return out, [left, right, top, down, oh, ow]

    def __call__(self, image):
        """[n,h,w,c] float32 -> [n,c,H,W] (FastCOCOTransform.__call__)."""
        return self._run(image, 0, 0.0)[0]

    def padded(self, image, size_divisor=32, pad_value=0):
        """transform(image) followed by pad(): one kernel.  Returns (image, pad_info) like infer.pad."""
        return self._run(image, size_divisor, pad_value)


def pad(image, size_divisor=32, pad_value=0):
    """infer.pad: centred padding of an NCHW tensor to a multiple of size_divisor.
    Returns (image, [left, right, top, down, new_height, new_width])."""
    _lib.require_cuda_tensor(image, "image", torch.float32)
    height, width = image.shape[-2:]
    new_height = int(math.ceil(height / size_divisor) * size_divisor)
    new_width = int(math.ceil(width / size_divisor) * size_divisor)
    pad_left, pad_top = (new_width - width) // 2, (new_height - height) // 2
    pad_right, pad_down = new_width - width - pad_left, new_height - height - pad_top
    pad_info = [pad_left, pad_right, pad_top, pad_down, new_height, new_width]
    if new_height == height and new_width == width:
        return image, pad_info
    src = image.contiguous()
    planes = src.numel() // (height * width)
    out = torch.empty(tuple(src.shape[:-2]) + (new_height, new_width), dtype=torch.float32, device=src.device)
    L = _lib.load()
    with torch.cuda.device(src.device):
        rc = L.om_pad_nchw(ctypes.c_void_p(src.data_ptr()), planes, height, width, pad_top, pad_left, new_height,
                           new_width, float(pad_value), ctypes.c_void_p(out.data_ptr()),
                           _lib.current_stream_ptr(src.device))
    _lib.check(rc, "om_pad_nchw")
    return out, pad_info


def build_transform(config):
    """trainer/builder.py:108-115: config = dict(type='FastCOCOTransform', pipeline=[dict(type='Resize', ...), ...])."""
    import sys
    kwargs = dict(config)
    cls = getattr(sys.modules[__name__], kwargs.pop("type"))
    items = []
    for item in kwargs.pop("pipeline"):
        it = dict(item)
        items.append(getattr(cls, it.pop("type"))(**it))
    return cls(pipeline=items, **kwargs)
