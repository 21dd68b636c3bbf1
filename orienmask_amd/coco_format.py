"""COCO-format conversion of detections on the GPU.

Mirrors COCOMetrics.to_coco_format (/root/reference/eval/coco_eval.py:57-63,108-145): boxes are mapped back
to the original image (`_recover_shape_bbox`, :146-189), masks are un-padded, flipped, resized to the original
size and rounded (`_recover_shape_segm`, :191-205) and run-length encoded like
``maskUtils.encode(np.asfortranarray(mask))`` (:120-122).  The resize and the RLE run on the device
(``om_recover_bbox`` / ``om_recover_masks_rle``); only the run lengths are copied to the host, where they are
packed into pycocotools' string form.  pycocotools itself is not available offline: the string packing restates
its published ``rleToString`` and is unpinned; boxes, resized masks and run lengths are pinned.

Round 4: ``COCOFormatter.to_coco_format`` converts a whole batch with the strings packed ON THE DEVICE
(``om_recover_masks_rle_strings``): every image's kernels are enqueued without a host read, all masks append their
strings to one byte buffer, and the host makes three copies per batch (boxes + scores + classes; string offsets;
the used part of the byte buffer).  ``rle_to_string`` below stays as the slow path of a mask whose runs or string
overflow the first-guess buffers, and as the single-mask helper ``recover_masks_rle`` uses.
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


def _rle_image(mask_u8, sample_info):
    """om_rle_image of one image's [K,H,W] uint8 masks."""
    top, down, left, right = _crop_of(sample_info)
    return _lib.RleImage(mask_u8.data_ptr(), mask_u8.shape[0], mask_u8.shape[1], mask_u8.shape[2], top, down, left, right,
                         int(bool(sample_info.get("hflip", False))), int(bool(sample_info.get("vflip", False))),
                         int(sample_info["height"]), int(sample_info["width"]))


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

    MAX_RUNS = 8192                 # first-guess run buffer per mask (a 544^2 orientation mask has a few hundred runs)
    BYTES_PER_MASK = 4096           # first-guess share of the batch's string buffer per mask

    WORST_CASE_BYTES = 256 << 20    # most scratch one worst-case launch may take: overflowing masks are redone in chunks of this size

    def __init__(self, cat2label, with_mask=True):
        self.cat2label = list(cat2label)
        self.with_mask = with_mask
        self._scratch = {}              # (device, name) -> tensor: the batch's run / string / header buffers, grown on demand

    def _buffer(self, dev, name, numel, dtype):
        """A scratch tensor of at least `numel` elements (the evaluation loop formats hundreds of batches: ~36 KB per detection
        would otherwise be allocated and freed per batch).  Grows geometrically; contents are undefined."""
        key = (str(dev), name)
        t = self._scratch.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            grown = max(numel, int(t.numel() * 1.5) if t is not None and t.dtype == dtype else 0)
            t = torch.empty(grown, dtype=dtype, device=dev)
            self._scratch[key] = t
        return t[:numel]

    def _worst_case_strings(self, items):
        """[(mask [1,H,W], info)] -> strings, with buffers for the worst case (every pixel its own run, six characters per
        run): one launch and two host reads per chunk of at most WORST_CASE_BYTES of scratch (~3 MB per 480 x 640 mask)."""
        per_mask = 10 * (max(int(i["height"]) * int(i["width"]) for _, i in items) + 1)     # 4 B of counts + 6 B of string per run
        step = max(1, self.WORST_CASE_BYTES // per_mask)
        out = []
        for i0 in range(0, len(items), step):
            out += self._worst_case_chunk(items[i0:i0 + step])
        return out

    @staticmethod
    def _worst_case_chunk(items):
        L = _lib.load()
        dev = items[0][0].device
        n = len(items)
        full = max(int(i["height"]) * int(i["width"]) for _, i in items) + 1
        cap = min(6 * full * n, (1 << 31) - 2)
        counts = torch.empty((n, full), dtype=torch.int32, device=dev)
        hdr = torch.zeros(2 + 3 * n, dtype=torch.int32, device=dev)            # cursor, overflow | off | len | n_runs
        sbytes = torch.empty(cap, dtype=torch.uint8, device=dev)
        masks = [m.contiguous().view(torch.uint8) for m, _ in items]
        imgs = (_lib.RleImage * n)(*[_rle_image(m, i) for m, (_, i) in zip(masks, items)])
        with torch.cuda.device(dev):
            _lib.check(L.om_recover_masks_rle_strings(
                imgs, n, ctypes.c_void_p(counts.data_ptr()), full, ctypes.c_void_p(hdr.data_ptr() + (2 + 2 * n) * 4),
                ctypes.c_void_p(sbytes.data_ptr()), cap, ctypes.c_void_p(hdr.data_ptr()), ctypes.c_void_p(hdr.data_ptr() + 8),
                ctypes.c_void_p(hdr.data_ptr() + (2 + n) * 4), _lib.current_stream_ptr(dev)), "om_recover_masks_rle_strings")
        h = hdr.cpu().tolist()
        if h[1] or any(o < 0 for o in h[2:2 + n]):
            raise _lib.OrienMaskHipError("RLE strings of %d masks do not fit %d bytes" % (n, cap))
        raw = bytes(sbytes[:h[0]].cpu().numpy())
        return [raw[o:o + ln].decode("ascii") for o, ln in zip(h[2:2 + n], h[2 + n:2 + 2 * n])]

    def to_coco_format(self, batch_info, detections):
        """/root/reference/eval/coco_eval.py:57-63 for one batch.  Nothing is read back until every image's kernels are
        enqueued; then boxes / scores / classes come in one copy, the strings in two (offsets, bytes)."""
        pairs = [(info, det) for info, det in zip(batch_info, detections) if det["bbox"].numel() != 0]
        out = {"bbox": []}
        if self.with_mask:
            out["segm"] = []
        if not pairs:
            return out
        dev = pairs[0][1]["bbox"].device
        _lib.require_cuda_tensor(pairs[0][1]["bbox"], "bbox", torch.float32)
        L = _lib.load()
        Ks = [int(det["bbox"].shape[0]) for _, det in pairs]
        N = sum(Ks)
        starts = [0]
        for k in Ks:
            starts.append(starts[-1] + k)
        xywh = torch.empty((N, 4), dtype=torch.float32, device=dev)
        st = _lib.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            for (info, det), s0, K in zip(pairs, starts, Ks):
                b = det["bbox"].contiguous()
                cp, pd = info.get("collate_pad"), info.get("pad")
                cp_arr = (ctypes.c_int32 * 6)(*[int(v) for v in cp]) if cp is not None else None
                pd_arr = (ctypes.c_int32 * 6)(*[int(v) for v in pd]) if pd is not None else None
                _lib.check(L.om_recover_bbox(ctypes.c_void_p(b.data_ptr()), K, b.shape[1], cp_arr, pd_arr,
                                             int(bool(info.get("hflip", False))), int(bool(info.get("vflip", False))),
                                             int(info["height"]), int(info["width"]),
                                             ctypes.c_void_p(xywh.data_ptr() + s0 * 16), st), "om_recover_bbox")
            if self.with_mask:
                cap = N * self.BYTES_PER_MASK
                hdr = self._buffer(dev, "hdr", 2 + 3 * N, torch.int32)        # cursor, overflow | off[N] | len[N] | n_runs[N]
                hdr.zero_()
                sbytes = self._buffer(dev, "strings", cap, torch.uint8)
                counts = self._buffer(dev, "counts", N * self.MAX_RUNS, torch.int32).view(N, self.MAX_RUNS)
                masks = []                                                      # keep the uint8 views alive until the launch
                imgs = (_lib.RleImage * len(pairs))()
                for i, (info, det) in enumerate(pairs):
                    _lib.require_cuda_tensor(det["mask"], "mask")
                    masks.append(det["mask"].contiguous().view(torch.uint8))
                    imgs[i] = _rle_image(masks[-1], info)
                _lib.check(L.om_recover_masks_rle_strings(
                    imgs, len(pairs), ctypes.c_void_p(counts.data_ptr()), self.MAX_RUNS, ctypes.c_void_p(hdr.data_ptr() + (2 + 2 * N) * 4),
                    ctypes.c_void_p(sbytes.data_ptr()), cap, ctypes.c_void_p(hdr.data_ptr()), ctypes.c_void_p(hdr.data_ptr() + 8),
                    ctypes.c_void_p(hdr.data_ptr() + (2 + N) * 4), st), "om_recover_masks_rle_strings")
        # ---- the batch's host reads
        meta = torch.cat([xywh, torch.cat([det["bbox"][:, -1:] for _, det in pairs]),
                          torch.cat([det["cls"].reshape(-1, 1).to(torch.float32) for _, det in pairs])], dim=1).cpu()
        boxes = meta[:, :4].tolist()
        scores = meta[:, 4].tolist()
        cats = [self.cat2label[int(c)] for c in meta[:, 5].tolist()]
        ids = [info["id"] for (info, _), K in zip(pairs, Ks) for _ in range(K)]
        out["bbox"] = [{"image_id": i, "category_id": c, "bbox": b, "score": s} for i, c, b, s in zip(ids, cats, boxes, scores)]
        if self.with_mask:
            h = hdr.cpu().tolist()
            used = min(h[0], cap)
            raw = bytes(sbytes[:used].cpu().numpy()) if used > 0 else b""
            offs, lens = h[2:2 + N], h[2 + N:2 + 2 * N]
            sizes = [[int(info["height"]), int(info["width"])] for (info, _), K in zip(pairs, Ks) for _ in range(K)]
            strings = [raw[o:o + n].decode("ascii") if o >= 0 else None for o, n in zip(offs, lens)]
            if any(s is None for s in strings):
                # masks with more runs than MAX_RUNS, or a batch whose strings outgrew the buffer: those masks again, in one
                # launch, with worst-case buffers (still packed on the device)
                todo = [(s0 + k, det["mask"][k:k + 1], info) for (info, det), s0, K in zip(pairs, starts, Ks) for k in range(K)
                        if strings[s0 + k] is None]
                for (idx, _, _), text in zip(todo, self._worst_case_strings([(m, i) for _, m, i in todo])):
                    strings[idx] = text
            out["segm"] = [{"image_id": i, "category_id": c, "segmentation": {"size": sz, "counts": cs}, "score": s}
                           for i, c, sz, cs, s in zip(ids, cats, sizes, strings, scores)]
        return out
