"""CPU: pin the oracle against fixtures generated from the real reference.

tests/golden/*.npz were written by tools/gen_golden.py, which imports /root/reference (model,
postprocess, and the reference's own nms_cpu.cpp compiled into oracle/_ref).  Nothing here
reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, fixture_weights_and_input, golden_files, post_cfg
from oracle import orienmask_ref as R
from orienmask_amd import synth


def unpack_masks(packed, shape):
    shape = tuple(int(s) for s in shape)
    if shape[0] == 0:
        return np.zeros(shape, dtype=bool)
    bits = np.unpackbits(packed, axis=1)[:, :shape[1] * shape[2]]
    return bits.reshape(shape).astype(bool)


def _post_oracle(size):
    pc = post_cfg(size)
    return R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], pc["num_classes"],
                               conf_thresh=pc["conf_thresh"], nms_thresh=0.5, nms_pre=pc["nms_pre"],
                               nms_post=pc["nms_post"], orien_thresh=pc["orien_thresh"]), pc


def test_nms_known_answers():
    kat = np.load(os.path.join(GOLDEN, "nms_kat.npz"))
    names = sorted(k[:-5] for k in kat.files if k.endswith("_keep"))
    assert len(names) >= 12
    for name in names:
        dets = torch.from_numpy(kat[name + "_dets"]); cats = torch.from_numpy(kat[name + "_cats"])
        thr = float(kat[name + "_thr"])
        _, _, keep = R.batched_nms(dets, cats, thr)
        assert keep.numpy().tolist() == kat[name + "_keep"].tolist(), name
        plain = R.nms_cpu(dets, thr)
        assert plain.numpy().tolist() == kat[name + "_keep_plain"].tolist(), name
        # the numpy restatement agrees with the C one
        assert R.nms_numpy(dets.numpy(), thr).tolist() == plain.numpy().tolist(), name


def _nms_cuda_numpy(dets, thr):
    """Independent numpy float32 restatement of nms_kernel.cu (devIoU :13-23, strict > :62, host loop :123-134, order_t[keep]
    :136-139) without the 64-wide tiling: greedy over the stable score-descending order."""
    d = np.asarray(dets, dtype=np.float32)
    n = d.shape[0]
    order = torch.sort(torch.from_numpy(d[:, 4].copy()), stable=True, dim=0, descending=True)[1].numpy()
    s = d[order]
    two = np.float32(2)
    l, r = s[:, 0] - s[:, 2] / two, s[:, 0] + s[:, 2] / two
    t, b = s[:, 1] - s[:, 3] / two, s[:, 1] + s[:, 3] / two
    area = s[:, 2] * s[:, 3]
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        js = np.arange(i + 1, n)
        w = np.maximum(np.minimum(r[i], r[js]) - np.maximum(l[i], l[js]), np.float32(0))
        h = np.maximum(np.minimum(b[i], b[js]) - np.maximum(t[i], t[js]), np.float32(0))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[js] - inter)
        removed[js[iou > np.float32(thr)]] = True
    return order[np.array(keep, dtype=np.int64)]


def test_nms_cuda_backend_oracle():
    """oracle/nms_cuda_ref.c (the reference's CUDA backend, nms_kernel.cu:13-140, which cannot be built: parity of this
    backend is pinned by restatement only) against the independent numpy restatement above, and the semantic differences
    from the CPU backend on hand-built boxes: IoU == threshold is kept (strict >), keep comes back score-descending."""
    d = torch.tensor([[1.5, 0.5, 3.0, 1.0, 0.9], [2.5, 0.5, 3.0, 1.0, 0.8], [8.0, 0.5, 3.0, 1.0, 0.95]])
    assert R.nms_cpu(d, 0.5).tolist() == [0, 2] and R.nms_cuda(d, 0.5).tolist() == [2, 0, 1]
    assert R.nms_cuda(d, 0.4999999).tolist() == [2, 0]
    assert R.nms_cuda(torch.zeros(0, 5), 0.5).tolist() == []
    kat = np.load(os.path.join(GOLDEN, "nms_kat.npz"))
    for name in sorted(k[:-5] for k in kat.files if k.endswith("_keep")):
        dets = torch.from_numpy(kat[name + "_dets"])
        if dets.shape[0] == 0:
            continue
        thr = float(kat[name + "_thr"])
        assert R.nms_cuda(dets, thr).tolist() == _nms_cuda_numpy(dets.numpy(), thr).tolist(), name
    rng = np.random.Generator(np.random.PCG64(5))
    for n in (63, 64, 65, 129, 700):
        dn = np.concatenate([rng.random((n, 2)), rng.random((n, 2)) * 0.3 + 0.02, rng.random((n, 1))], 1).astype(np.float32)
        dn[::5, 4] = dn[1::5, 4][:dn[::5].shape[0]]                       # score ties: visited in index order
        assert R.nms_cuda(torch.from_numpy(dn), 0.45).tolist() == _nms_cuda_numpy(dn, 0.45).tolist(), n


@pytest.mark.parametrize("fname", golden_files("post_"))
def test_postprocess_matches_reference_bit_exact(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    post, pc = _post_oracle(size)
    heads = synth.synth_heads(int(g["seed"]), batch, pc["grid_size"], regime=str(g["regime"]))
    res = post(heads)
    assert len(res) == batch
    for b, r in enumerate(res):
        assert np.array_equal(r["bbox"].numpy(), g["bbox%d" % b]), (fname, b)
        assert np.array_equal(r["cls"].numpy(), g["cls%d" % b]), (fname, b)
        want = unpack_masks(g["mask%d" % b], g["maskshape%d" % b])
        assert np.array_equal(r["mask"].numpy(), want), (fname, b)


@pytest.mark.parametrize("fname", golden_files("fwd_"))
def test_forward_matches_reference(fname):
    g = np.load(os.path.join(GOLDEN, fname))
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    if size[0] > 200 and os.environ.get("OM_FAST_TESTS"):
        pytest.skip("544x544 CPU forward skipped under OM_FAST_TESTS")
    sd, x = fixture_weights_and_input(g)
    out, feats = R.forward(sd, x, return_features=True)
    tensors = dict(bbox32=out[0][0], bbox16=out[1][0], bbox8=out[2][0],
                   oriens=torch.cat([out[0][1], out[1][1], out[2][1]], 1),
                   x32=feats["x32"], x16=feats["x16"], x8=feats["x8"], x4=feats["x4"])
    for k, t in tensors.items():
        assert list(t.shape) == g[k + "_shape"].tolist(), k
        flat = t.reshape(-1)
        samp = flat[torch.from_numpy(g[k + "_idx"])].numpy()
        # same torch CPU primitives, same layouts -> identical bits
        assert np.array_equal(samp, g[k + "_samples"]), k
        assert abs(flat.double().sum().item() - g[k + "_sum"][0]) <= 1e-9 * max(1.0, g[k + "_sum"][1]), k
        if k in g.files:
            assert np.array_equal(t.numpy(), g[k]), k
    post, _ = _post_oracle(size)
    for b, r in enumerate(post(out)):
        assert np.array_equal(r["bbox"].numpy(), g["bbox_det%d" % b])
        assert np.array_equal(r["cls"].numpy(), g["cls_det%d" % b])
        assert np.array_equal(r["mask"].numpy(), unpack_masks(g["mask%d" % b], g["maskshape%d" % b]))


def test_bilinear_restatement_matches_torch():
    """The HIP mask kernel's evaluation order vs F.interpolate: bit-identical with this torch
    build's CPU kernel; never more than a few ulps on any other."""
    rng = np.random.Generator(np.random.PCG64(5))
    p = rng.standard_normal((24, 40), dtype=np.float32)
    want = torch.nn.functional.interpolate(torch.from_numpy(p)[None, None], scale_factor=4.0, mode="bilinear",
                                           align_corners=False)[0, 0].numpy()
    got = R.bilinear_x4_restated(p)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 4 * np.finfo(np.float32).eps * np.max(np.abs(p))
    mism = int((got != want).sum())
    if mism:
        import warnings
        warnings.warn("bilinear restatement differs from torch in %d of %d pixels (<= 4 ulp)" % (mism, want.size))


def test_yolo_forward_matches_reference():
    """The non-Plus OrienMaskYOLO restatement vs the reference's own model."""
    g = np.load(os.path.join(GOLDEN, "yolo_fwd.npz"))
    for name in ("y96_b2", "y128x160_b1"):
        wseed, xseed, batch, h, w = (int(v) for v in g[name + "_meta"])
        sd = synth.synth_state_dict(wseed, obj_bias=-16.0, head_gain=4.0, model="OrienMaskYOLO")
        assert len(sd) == 506
        out = R.forward_yolo(sd, synth.synth_image_batch(xseed, batch, h, w))
        got = dict(bbox32=out[0][0], bbox16=out[1][0], bbox8=out[2][0],
                   oriens=torch.cat([out[0][1], out[1][1], out[2][1]], 1))
        for k, t in got.items():
            assert np.array_equal(t.numpy(), g["%s_%s" % (name, k)]), (name, k)


def test_preprocess_matches_reference():
    """FastCOCOTransform (Resize + Normalize) + infer.pad, generated from the reference's own code."""
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    names = sorted(k[:-5] for k in g.files if k.endswith("_seed"))
    assert len(names) == 5
    for name in names:
        n, h, w = (int(v) for v in g[name + "_shape"])
        size = tuple(int(v) for v in g[name + "_size"])
        img = synth.synth_photo_batch(int(g[name + "_seed"]), n, h, w)
        x = R.fast_coco_transform(img, size if size != (h, w) else None)
        x, info = R.pad_to_divisor(x, 32, 0)
        assert info == g[name + "_pad"].tolist(), name
        assert list(x.shape) == g[name + "_outshape"].tolist(), name
        flat = x.reshape(-1)
        assert np.array_equal(flat[torch.from_numpy(g[name + "_idx"])].numpy(), g[name + "_samples"]), name
        if name + "_out" in g.files:
            assert np.array_equal(x.numpy(), g[name + "_out"]), name


def _coco_cases():
    import json
    g = np.load(os.path.join(GOLDEN, "coco_format.npz"))
    for name in sorted(k[:-3] for k in g.files if k.endswith("_hw")):
        H, W = (int(v) for v in g[name + "_hw"])
        info = json.loads(str(g[name + "_info"]))
        masks = unpack_masks(g[name + "_mask"], (g[name + "_bbox"].shape[0], H, W))
        seg = unpack_masks(g[name + "_seg"], g[name + "_segshape"])
        yield name, info, masks, g[name + "_bbox"], g[name + "_xywh"], seg


def test_coco_format_recover_matches_reference():
    """_recover_shape_bbox / _recover_shape_segm restatements vs the reference's static methods; RLE properties."""
    n = 0
    for name, info, masks, bbox, xywh, seg in _coco_cases():
        assert np.array_equal(R.recover_shape_bbox(torch.from_numpy(bbox), info).numpy(), xywh), name
        got = R.recover_shape_segm(torch.from_numpy(masks), info).numpy()
        assert np.array_equal(got.astype(bool), seg), name
        for k in range(got.shape[0]):
            c = R.rle_counts(got[k])
            assert sum(c) == got[k].size and sum(c[1::2]) == int(got[k].sum())
            s = R.rle_to_string(c)
            assert R.rle_string_decode(s, got[k].size) == c          # string packing round-trips
        n += 1
    assert n == 5
