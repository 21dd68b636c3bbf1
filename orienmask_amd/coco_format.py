"""COCO-format conversion of detections on the GPU.

Mirrors COCOMetrics.to_coco_format (/root/reference/eval/coco_eval.py:57-63,108-145): boxes are mapped back
to the original image (`_recover_shape_bbox`, :146-189), masks are un-padded, flipped, resized to the original
size and rounded (`_recover_shape_segm`, :191-205) and run-length encoded like
``maskUtils.encode(np.asfortranarray(mask))`` (:120-122).  The resize and the RLE run on the device
(``om_recover_bbox`` / ``om_recover_masks_rle``); only the run lengths are copied to the host, where they are
packed into pycocotools' string form.  pycocotools itself is not available offline: the string packing restates
its published ``rleToString`` and is unpinned; boxes, resized masks and run lengths are pinned.
"""
import ctypes

import torch

from . import lib as _lib


def rle_to_string(counts):
    """pycocotools rleToString (maskApi.c): differences against the run two back, 5 data bits per char,
    bit 5 = continuation, offset 48."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def _crop_of(sample_info):
    top = down = left = right = 0
    if sample_info.get("collate_pad") is not None:
        l, r, t, d = sample_info["collate_pad"][:4]
        left += l; right += r; top += t; down += d
    if sample_info.get("pad") is not None:
        t, d, l, r = sample_info["pad"][:4]
        left += l; right += r; top += t; down += d
    return top, down, left, right


def recover_shape_bbox(bbox, sample_info):
    """[K,>=4] normalised (cx,cy,w,h,..) CUDA tensor -> [K,4] x,y,w,h in original pixels (device)."""
    _lib.require_cuda_tensor(bbox, "bbox", torch.float32)
    K = bbox.shape[0]
    out = torch.empty((K, 4), dtype=torch.float32, device=bbox.device)
    if K == 0:
        return out
    b = bbox.contiguous()
    cp = sample_info.get("collate_pad")
    pd = sample_info.get("pad")
    cp_arr = (ctypes.c_int32 * 6)(*[int(v) for v in cp]) if cp is not None else None
    pd_arr = (ctypes.c_int32 * 6)(*[int(v) for v in pd]) if pd is not None else None
    with torch.cuda.device(b.device):
        rc = _lib.load().om_recover_bbox(ctypes.c_void_p(b.data_ptr()), K, b.shape[1], cp_arr, pd_arr,
                                         int(bool(sample_info.get("hflip", False))), int(bool(sample_info.get("vflip", False))),
                                         int(sample_info["height"]), int(sample_info["width"]),
                                         ctypes.c_void_p(out.data_ptr()), _lib.current_stream_ptr(b.device))
    _lib.check(rc, "om_recover_bbox")
    return out


def recover_masks_rle(mask, sample_info, max_runs=None, return_resized=False):
    """[K,H,W] bool/uint8 CUDA tensor -> list of K COCO RLE dicts {'size': [h, w], 'counts': str}
    (and optionally the resized uint8 masks [K,h,w], what _recover_shape_segm returns)."""
    _lib.require_cuda_tensor(mask, "mask")
    K, H, W = mask.shape
    oh, ow = int(sample_info["height"]), int(sample_info["width"])
    resized = torch.empty((K, oh, ow), dtype=torch.uint8, device=mask.device) if return_resized else None
    if K == 0:
        return ([], resized) if return_resized else []
    m = mask.contiguous().view(torch.uint8)
    top, down, left, right = _crop_of(sample_info)
    full = oh * ow + 1
    max_runs = min(full, max_runs or 16384)
    L = _lib.load()
    while True:
        counts = torch.empty((K, max_runs), dtype=torch.int32, device=mask.device)
        n_runs = torch.empty((K,), dtype=torch.int32, device=mask.device)
        with torch.cuda.device(mask.device):
            rc = L.om_recover_masks_rle(ctypes.c_void_p(m.data_ptr()), K, H, W, top, down, left, right,
                                        int(bool(sample_info.get("hflip", False))), int(bool(sample_info.get("vflip", False))),
                                        oh, ow, ctypes.c_void_p(counts.data_ptr()), max_runs,
                                        ctypes.c_void_p(n_runs.data_ptr()),
                                        ctypes.c_void_p(resized.data_ptr()) if return_resized else None,
                                        _lib.current_stream_ptr(mask.device))
        _lib.check(rc, "om_recover_masks_rle")
        n = n_runs.cpu().tolist()
        if max(n) <= max_runs or max_runs >= full:
            break
        max_runs = full                      # a pathological mask: redo with the worst-case buffer
    host = counts[:, :max(n)].cpu()
    rles = [{"size": [oh, ow], "counts": rle_to_string((host[k, :n[k]].numpy().astype("int64") & 0xFFFFFFFF).tolist())}
            for k in range(K)]
    return (rles, resized) if return_resized else rles


class COCOFormatter:
    """to_coco_format of COCOMetrics without pycocotools: same result dicts, computed on the device."""

    def __init__(self, cat2label, with_mask=True):
        self.cat2label = list(cat2label)
        self.with_mask = with_mask

    def to_coco_format(self, batch_info, detections):
        bbox_results, segm_results = [], []
        for info, det in zip(batch_info, detections):
            if det["bbox"].numel() == 0:
                continue
            scores = det["bbox"][:, -1].tolist()
            cats = [self.cat2label[c] for c in det["cls"].flatten().tolist()]
            xywh = recover_shape_bbox(det["bbox"], info).tolist()
            for box, score, cat in zip(xywh, scores, cats):
                bbox_results.append({"image_id": info["id"], "category_id": cat, "bbox": box, "score": score})
            if self.with_mask:
                for rle, score, cat in zip(recover_masks_rle(det["mask"], info), scores, cats):
                    segm_results.append({"image_id": info["id"], "category_id": cat, "segmentation": rle, "score": score})
        out = {"bbox": bbox_results}
        if self.with_mask:
            out["segm"] = segm_results
        return out
