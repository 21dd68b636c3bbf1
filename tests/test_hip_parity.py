"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (BASELINE.json north_star): class / index results bit-exact, box coordinates and head
tensors within 1e-4 (relative to the tensor's scale), mask IoU >= 1 - 1e-4.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, fixture_weights_and_input, golden_files, post_cfg
from oracle import orienmask_ref as R
from orienmask_amd import lib as omlib
from orienmask_amd import synth
from test_oracle_golden import unpack_masks

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4
_ORACLE_CACHE = {}       # CPU-oracle results shared by the parametrisations of one test (the GPU suite's wall time is mostly this)


@pytest.fixture(scope="module")
def dev(built):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    omlib.load()
    return torch.device("cuda:0")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _rel_err(got, want):
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


def _ulps(a, b):
    """Element-wise distance in float32 units in the last place (same-sign finite values)."""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def _check_detections(r, want_bbox, want_cls, want_mask, tag, exact_decode):
    """One image's detections against the reference's.  exact_decode: the heads fed to both sides were bit-identical,
    so scores and box centres must be bit-identical too (csrc/ref_math.h restates torch's sigmoid paths exactly) and the box
    sizes within 2 ulps (MKL's vsExp cannot be restated); otherwise the 1e-4 budget of north_star applies."""
    assert r["bbox"].shape[0] == want_bbox.shape[0], (tag, r["bbox"].shape, want_bbox.shape)
    assert r["mask"].dtype == torch.bool and r["cls"].dtype == torch.long
    assert np.array_equal(r["cls"].cpu().numpy(), want_cls), tag                  # indices: bit-exact
    got = r["bbox"].cpu().numpy()
    if want_bbox.shape[0]:
        assert np.max(np.abs(got - want_bbox)) <= 1e-4 * max(1.0, np.abs(want_bbox).max()), tag
        if exact_decode:
            assert np.array_equal(got[:, [0, 1, 4]], want_bbox[:, [0, 1, 4]]), (tag, "cx / cy / score not bit-identical")
            assert _ulps(got[:, 2:4], want_bbox[:, 2:4]).max() <= 2, (tag, "w / h off by more than 2 ulps")
    got_mask = r["mask"].cpu().numpy()
    assert got_mask.shape == want_mask.shape, tag
    for k in range(want_mask.shape[0]):
        assert _mask_iou(got_mask[k], want_mask[k]) >= 1 - 1e-4, (tag, k)


def _margin_ctx(oracle_post, heads_cpu, b):
    """What _check_detections_composed needs to judge a differing mask pixel: the orientation field of image b (the oracle's
    arithmetic on the heads under test) and, per detection of the oracle's postprocess ON THOSE SAME HEADS (bit-identical to the
    HIP postprocess, _check_detections), its box and anchor."""
    coord, score, cls, aidx, sel = oracle_post.candidates(heads_cpu, b)
    field = oracle_post.orien_field(heads_cpu, b)
    res = oracle_post.finish(coord, score, cls, aidx, field)
    return dict(field=field, bbox=res["bbox"].numpy(), cls=res["cls"].numpy(), anchor=res["anchor"].numpy(),
                grid_sizes=oracle_post.grid_sizes.numpy(), orien_thresh=oracle_post.orien_thresh)


def _borderline_flips(got_mask, want_mask, ctx, j, tol_rel=1e-4):
    """(number of differing pixels, the largest decision margin among them in units of the field's scale).  A pixel is inside a
    mask iff |Px - xc| < tx and |Py - yc| < ty (postprocess.py:157-164); its decision margin is how far the field would have to
    move to change that.  Two forwards that agree to 1e-6 of scale can only disagree on pixels whose margin is of that order."""
    diff = np.nonzero(got_mask.astype(bool) != want_mask.astype(bool))
    if diff[0].size == 0:
        return 0, 0.0
    a = int(ctx["anchor"][j]); gsx, gsy = ctx["grid_sizes"][a]
    cx, cy, w, h = (float(v) for v in ctx["bbox"][j, :4])
    f = ctx["field"][a].numpy()
    mx = np.abs(f[0][diff] - gsx * cx) - ctx["orien_thresh"] * w * gsx
    my = np.abs(f[1][diff] - gsy * cy) - ctx["orien_thresh"] * h * gsy
    inside = (mx < 0) & (my < 0)
    d = np.where(inside, np.minimum(-mx, -my), np.maximum(np.maximum(mx, 0), np.maximum(my, 0)))
    scale = max(float(np.abs(f).max()), 1.0)
    return int(diff[0].size), float(d.max() / scale)


FALLBACK_MARGIN, FALLBACK_PIXELS, FALLBACK_SHARE = 1e-6, 4, 0.02
_FALLBACK_TALLY = {}


def _check_detections_composed(r, want_bbox, want_cls, want_mask, tag, score_tol=5e-5, nms_post=100, margin_ctx=None):
    """HIP forward + HIP postprocess against the reference's forward + postprocess.  The two forwards differ by ~1e-6 of the
    head tensors' scale (another summation order), i.e. scores differ by up to a few 1e-5 relative, so the comparison is
    exact EXCEPT where the reference's own answer hinges on a gap smaller than that:
      * detections are matched one to one irrespective of position: same class, box within 1e-4, and mask IoU >= 1 - 1e-4
        (north_star).  ONE boundary pixel is already more than 1e-4 of a mask smaller than 10^4 pixels, so a mask that misses
        that bar may still pass through the FALLBACK: every differing pixel BORDERLINE -- its decision margin
        (_borderline_flips, evaluated in the oracle's arithmetic on the heads under test) at most FALLBACK_MARGIN = 1e-6 of the
        orientation field's scale, four times the largest margin ever observed (2.3e-7, profiles/r03_composed_flips.txt) -- and
        at most FALLBACK_PIXELS = 4 pixels of the mask differing (observed: at most 2).  How many masks took the fallback is
        counted per test -- per DISTINCT box: a candidate box is one detection per passing class, all with the same mask --
        printed, logged, and bounded: more than FALLBACK_SHARE = 2 % of a test's distinct boxes (and more than 2) fails (VERDICT
        round 5, task 5; rounds 3-5 allowed 1e-5 and 0.1 % + 2 pixels with no count).  Without margin_ctx only images under 200 pixels
        fall back to "IoU >= 0.999 or at most 2 pixels".  The observed flips are printed per image (pytest -s) and appended to
        gpurun_out/composed_flips.txt when that directory exists;
      * position by position the scores agree within score_tol: detections may only trade places with near-ties;
      * when the list is cut at nms_post, a detection within score_tol of the last score may be replaced by its runner-up.
    Exact ties inside the reference's list (torch.topk / sort leave their order unspecified) are covered by the same rule."""
    K = want_bbox.shape[0]
    assert r["bbox"].shape[0] == K, (tag, r["bbox"].shape, want_bbox.shape)
    if K == 0:
        return
    got_bbox = r["bbox"].cpu().numpy(); got_cls = r["cls"].cpu().numpy(); got_mask = r["mask"].cpu().numpy()
    ws, gs = want_bbox[:, 4], got_bbox[:, 4]
    if K >= 2 and (np.diff(ws) <= 0).all():                     # score-ordered output: positions may only move among near-ties
        assert np.all(np.abs(gs - ws) <= score_tol * np.maximum(ws, 1e-3)), (tag, np.abs(gs - ws).max())
    ctx_of = None
    if margin_ctx is not None:       # the context's detections are the HIP postprocess's own (same heads in -> same bits out;
        key = lambda bb, c: (np.ascontiguousarray(bb[[0, 1, 4]]).tobytes(), int(c))       # centres and score are bit-identical, w / h <= 2 ulps
        lut = {key(margin_ctx["bbox"][k], margin_ctx["cls"][k]): k for k in range(margin_ctx["bbox"].shape[0])}   # order may differ
        ctx_of = [lut.get(key(got_bbox[j], got_cls[j])) for j in range(K)]                                        # inside exact-tie groups)
        assert all(k is not None for k in ctx_of), (tag, "the oracle's postprocess on the same heads found other detections")
    small = min(want_mask.shape[1:]) < 200
    used = np.zeros(K, dtype=bool)
    unmatched, notes = [], []
    flips, worst_iou, worst_margin, most_flips, n_fallback = 0, 1.0, 0.0, 0, 0
    boxes, fallback_boxes = set(), set()        # distinct boxes: one candidate box is a detection per class that passes, with ONE mask
    for i in range(K):
        # box within 1e-4 of its own largest coordinate (>= 1: normalised units), _check_detections' measure per detection
        box_tol = 1e-4 * max(1.0, float(np.abs(want_bbox[i, :4]).max()))
        cand = np.nonzero((~used) & (got_cls == want_cls[i]) & (np.abs(got_bbox - want_bbox[i]).max(1) <= box_tol))[0]
        hit = None
        for j in cand:
            iou = _mask_iou(got_mask[j], want_mask[i])
            nflip = int(np.count_nonzero(got_mask[j].astype(bool) != want_mask[i].astype(bool)))
            margin = 0.0
            fell_back = False
            ok = iou >= 1 - 1e-4
            if not ok and margin_ctx is not None:
                nflip, margin = _borderline_flips(got_mask[j], want_mask[i], margin_ctx, ctx_of[j])
                ok = margin <= FALLBACK_MARGIN and nflip <= FALLBACK_PIXELS
                fell_back = ok
            elif not ok and small:
                ok = iou >= 0.999 or nflip <= 2
            if ok:
                hit = (j, iou, nflip, margin, fell_back)
                break
            notes.append((i, int(j), round(iou, 6), nflip, margin))
        if hit:
            used[hit[0]] = True
            flips += hit[2]
            worst_iou = min(worst_iou, hit[1])
            worst_margin = max(worst_margin, hit[3])
            most_flips = max(most_flips, hit[2])
            n_fallback += int(hit[4])
            boxes.add(want_bbox[i, :4].tobytes())
            if hit[4]:
                fallback_boxes.add(want_bbox[i, :4].tobytes())
        else:
            unmatched.append(i)
    # per test (PYTEST_CURRENT_TEST): distinct masks compared so far and how many of them needed the fallback.  Counted per distinct
    # BOX: a candidate box is one detection per class that passes (postprocess.py:102 lists (candidate, class) pairs) and all of
    # them carry the same mask, so one borderline pixel shows up in up to 80 detections
    tally = _FALLBACK_TALLY.setdefault(os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], [0, 0])
    tally[0] += len(boxes); tally[1] += len(fallback_boxes)
    line = ("composed %s: %d detections of %d distinct boxes, %d differing mask pixels in all (at most %d in one mask), worst mask IoU "
            "%.6f, largest decision margin of a differing pixel %.2e of the field's scale, %d masks of %d distinct boxes through the "
            "borderline fallback (this test so far: %d of %d distinct boxes), %d unmatched %s"
            % (tag, K, len(boxes), flips, most_flips, worst_iou, worst_margin, n_fallback, len(fallback_boxes), tally[1], tally[0],
               len(unmatched), notes[:6]))
    print(line)
    if os.path.isdir("gpurun_out"):
        with open(os.path.join("gpurun_out", "composed_flips.txt"), "a") as fh:
            fh.write(line + "\n")
    assert tally[1] <= max(2, FALLBACK_SHARE * tally[0]), (tag, "too many masks needed the borderline fallback", tally)
    at_cut = [i for i in unmatched if K == nms_post and ws[i] <= ws.min() * (1 + score_tol)]
    assert len(unmatched) == len(at_cut) <= 1, (tag, "detections without a counterpart", unmatched, ws[unmatched], notes[:6])
    # a candidate-ordered list (no top-k anywhere) has no freedom at all
    if not (K >= 2 and (np.diff(ws) <= 0).all()) and not unmatched:
        assert np.array_equal(got_cls, want_cls), tag


# ------------------------------------------------------------------------------------------------
# single convolutions
# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, W, cin, cout, k, stride, leaky, residual
    (2, 16, 16, 32, 64, 3, 2, 1, False),
    (2, 12, 20, 64, 32, 1, 1, 1, False),
    (1, 17, 17, 32, 64, 3, 1, 1, True),
    (3, 8, 8, 128, 128, 3, 1, 1, True),
    (2, 9, 7, 256, 255, 1, 1, 0, False),
    (1, 10, 10, 64, 18, 1, 1, 0, False),
    (2, 6, 6, 384, 128, 1, 1, 1, False),
    (1, 34, 34, 128, 256, 3, 1, 1, False),
    (5, 4, 4, 1024, 512, 1, 1, 1, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_layer_matches_torch(dev, case):
    B, H, W, cin, cout, k, stride, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    Ho, Wo = H // stride, W // stride
    res = torch.randn(B, cout, Ho, Wo, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, k // 2)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 31) // 32 * 32
    wp = torch.zeros(cpad, k * k * cin)
    wp[:cout] = w.permute(0, 2, 3, 1).reshape(cout, -1)
    sp = torch.zeros(cpad); sp[:cout] = scale
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, sd_, hd = wp.to(dev), sp.to(dev), hp.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    out = torch.full((B, Ho, Wo, cout), float("nan"), device=dev)
    rc = L.om_conv2d(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, k, stride, leaky,
                     _p(rd) if use_res else None, cout if use_res else 0, _p(out), cout,
                     omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d")
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert _rel_err(got, want) < 2e-6, case


@pytest.mark.parametrize("up,cin,cout,hw,cstride,coff", [(2, 512, 256, (5, 7), 768, 0), (8, 512, 64, (3, 3), 256, 64),
                                                       (4, 256, 64, (6, 5), 256, 128)])
def test_conv_upsample_epilogue_matches_torch(dev, up, cin, cout, hw, cstride, coff):
    """NearestUpsample fused into the producing 1x1 convolution (model/base.py:95-101, fpnplus.py:78-86): every output pixel
    is written up x up times into its channel slice of the concat buffer; the other channels of the buffer stay untouched."""
    L = omlib.load()
    H, W = hw
    B = 2
    g = torch.Generator().manual_seed(up * 100 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    want = torch.nn.functional.conv2d(x.double(), w.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.1)
    want = torch.nn.functional.interpolate(want, scale_factor=float(up), mode="nearest")
    cpad = (cout + 31) // 32 * 32
    wp = torch.zeros(cpad, cin); wp[:cout] = w.reshape(cout, cin)
    sp = torch.zeros(cpad); sp[:cout] = scale
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, sd_, hd = wp.to(dev), sp.to(dev), hp.to(dev)
    buf = torch.full((B, H * up, W * up, cstride), 7.0, device=dev)          # the concat buffer
    view = buf[..., coff:]
    rc = L.om_conv2d_mode(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, 1, 1, 1, None, 0,
                          ctypes.c_void_p(view.data_ptr()), cstride, 1, up, omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_mode")
    got = buf[..., coff:coff + cout].cpu().permute(0, 3, 1, 2).double()
    assert _rel_err(got, want) < 2e-6
    rest = torch.cat([buf[..., :coff], buf[..., coff + cout:]], -1)
    assert (rest == 7.0).all()


SPLIT_CONV_CASES = CONV_CASES + [
    (8, 68, 68, 256, 128, 1, 1, 1, False),     # 128 x 128 tiles
    (16, 68, 68, 128, 256, 3, 2, 1, False),    # 256 x 128 tiles, stride 2: padding on the top / left taps
    (4, 136, 136, 64, 32, 1, 1, 1, True),      # 128 x 32 tiles, residual
]


@pytest.mark.parametrize("case", SPLIT_CONV_CASES)
def test_conv_split_layer_matches_torch(dev, case):
    """conv_igemm_split.hip (fp32 activations split into hi/lo fp16 pairs in registers, packed hi/lo weights, three fp16 MFMAs
    per product group, fp32 accumulate) vs float64 convolution: same bound as the fp32-operand kernel."""
    from orienmask_amd.pack import conv_weights_split
    B, H, W, cin, cout, k, stride, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    Ho, Wo = H // stride, W // stride
    res = torch.randn(B, cout, Ho, Wo, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, k // 2)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 31) // 32 * 32
    ws, e = conv_weights_split(w, cpad)
    sp = torch.zeros(cpad); sp[:cout] = scale
    sp = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float()
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, sd_, hd = ws.to(dev), sp.to(dev), hp.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    out = torch.full((B, Ho, Wo, cout), float("nan"), device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = L.om_conv2d_split(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, k, stride, leaky,
                           _p(rd) if use_res else None, cout if use_res else 0, _p(out), cout, 0, 1, 0, 0, _p(status),
                           omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_split")
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert int(status.item()) == 0                      # finite outputs: the range guard stays clear
    err = _rel_err(got, want)
    print("conv split %s: %.2e" % (case, err))
    assert err < 2e-6, case


@pytest.mark.parametrize("case", [
    # (B, H, W, cin, cout, k, stride, leaky, residual, out_mode, tile, max_parts)
    (1, 17, 17, 512, 1024, 3, 1, 1, 0, 0, (64, 64), 8),      # backbone.conv6.x.conv.1 of one image: 80 tiles, 288 k-steps -> 6 parts
    (1, 17, 17, 1024, 512, 1, 1, 1, 0, 0, (64, 64), 8),      # conv6.x.conv.0: 40 tiles, 64 k-steps -> 4 parts
    (1, 34, 34, 256, 512, 3, 1, 1, 1, 0, (64, 64), 3),       # residual: the epilogue with loads; 152 tiles -> 3 parts
    (2, 34, 34, 256, 512, 3, 2, 1, 0, 0, (64, 64), 8),       # stride 2 (padding taps inside a part's first step), partial last M tile
    (1, 68, 68, 128, 256, 3, 1, 1, 1, 0, (128, 64), 8),      # the five-stage 128 x 64 form: 148 tiles -> 3 parts
    (1, 20, 12, 96, 255, 3, 1, 0, 0, 2, (64, 64), 8),        # head: ragged cout, NCHW output, 54 k-steps -> 3 parts of 18
    (1, 8, 8, 16, 64, 1, 1, 1, 0, 0, (64, 64), 8),           # one k-step: not cut
    (3, 17, 17, 96, 128, 3, 1, 1, 0, 0, (64, 64), 2),        # 54 k-steps: parts of 27 that start inside a kernel tap
])
def test_conv_split_k_parts(dev, case):
    """conv_igemm_split_kernel's split-K form (latency mode, om_model_set_latency_ksplit: a tile's k loop cut into parts, the last
    arrival sums the published accumulators in part order) against float64 convolution at the whole-tile kernel's bound, and twice
    in a row bit for bit (the sum order does not depend on which part arrives last)."""
    from orienmask_amd.pack import conv_weights_split
    B, H, W, cin, cout, k, stride, leaky, use_res, out_mode, (bm, bn), parts = case
    L = omlib.load()
    g = torch.Generator().manual_seed(B + H + W + cin + cout + k + parts)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    Ho, Wo = H // stride, W // stride
    res = torch.randn(B, cout, Ho, Wo, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, k // 2)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 31) // 32 * 32
    ws, e = conv_weights_split(w, cpad)
    sp = torch.zeros(cpad); sp[:cout] = scale
    sp = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float()
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, sd_, hd = ws.to(dev), sp.to(dev), hp.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(max_parts):
        shape = (B, cout, Ho, Wo) if out_mode == 2 else (B, Ho, Wo, cout)
        out = torch.full(shape, float("nan"), device=dev)
        rc = L.om_conv2d_split_k(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, k, stride, leaky,
                                 _p(rd) if use_res else None, cout if use_res else 0, _p(out), cout, out_mode, 1, bm, bn, max_parts,
                                 _p(status), omlib.current_stream_ptr(dev))
        omlib.check(rc, "om_conv2d_split_k")
        o = out.cpu()
        return o if out_mode == 2 else o.permute(0, 3, 1, 2)

    got = run(parts)
    assert torch.isfinite(got).all()
    assert int(status.item()) == 0
    err = _rel_err(got.double(), want)
    print("conv split-K %s: %.2e" % (case, err))
    assert err < 2e-6, case
    for _ in range(3):
        assert torch.equal(run(parts), got)
    whole = run(1)                                       # the same kernel with whole tiles
    assert _rel_err(whole.double(), want) < 2e-6
    assert _rel_err(got.double(), whole.double()) < 4e-6
    # other data through the same partial-tile area: nothing of the previous launch's parts may be read (a stale line in this
    # XCD's L2 or this CU's L1 would be exactly that)
    for rep in range(3):
        x2 = torch.randn(B, cin, H, W, generator=g)
        want2 = torch.nn.functional.conv2d(x2.double(), w.double(), None, stride, k // 2)
        want2 = want2 * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        if leaky:
            want2 = torch.where(want2 > 0, want2, want2 * 0.1)
        if use_res:
            want2 = want2 + res.double()
        xd.copy_(x2.permute(0, 2, 3, 1))
        assert _rel_err(run(parts).double(), want2) < 2e-6, (case, rep)


@pytest.mark.parametrize("case", [(8, 68, 68, 256, 128, 1, 1), (4, 68, 68, 128, 256, 3, 2), (2, 34, 34, 96, 128, 3, 2)])
def test_conv_split_wide_rows_equal_narrow(dev, case):
    """conv_igemm_split_wide_kernel (128 x 128 tile, ring stages of 128-byte operand rows: what the tile chooser's 128 x 128 runs
    where cin % 32 == 0) against conv_igemm_split_kernel<128,128> (64-byte rows; forced through the per-call tile shape): the same
    products in the same order per accumulator -- bit-identical outputs, stride-2 padding taps and a partial last M tile included."""
    from orienmask_amd.pack import conv_weights_split
    B, H, W, cin, cout, k, stride = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ws, e = conv_weights_split(w, cout)
    wd = ws.to(dev)
    sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
    hp = (torch.randn(cout, generator=g) * 0.2).to(dev)
    Ho, Wo = H // stride, W // stride
    outs = []
    for bm, bn in ((0, 0), (128, 128)):
        out = torch.full((B, Ho, Wo, cout), float("nan"), device=dev)
        omlib.check(L.om_conv2d_split(_p(x), B, H, W, cin, cin, _p(wd), _p(sp), _p(hp), cout, k, stride, 1, None, 0, _p(out), cout, 0, 1,
                                      bm, bn, None, omlib.current_stream_ptr(dev)), "om_conv2d_split")
        outs.append(out.cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("case", [
    # (B, H, W, cout, [(channels, up, pixel stride), ...]): neck4.0's four slices; neck16.0's route + feature (the feature a slice of
    # a wider buffer); one segment; a partial last M tile with three segments
    (2, 136, 136, 128, [(64, 8, 64), (64, 4, 64), (64, 2, 64), (64, 1, 256)]),
    (3, 34, 34, 256, [(256, 2, 256), (512, 1, 768)]),
    (1, 16, 16, 128, [(32, 4, 32)]),
    (5, 12, 20, 128, [(32, 1, 32), (96, 4, 96), (32, 2, 48)]),
])
def test_conv_split_gather_equals_materialised_concat(dev, case):
    """om_conv2d_split_gather (the 1x1 layer reading up-sampled slices where their producers stored them: what om_forward runs for
    neck16.0 / neck8.0 / neck4.0 in split-operand mode) against om_conv2d_split over torch's nearest up-sample + cat of the same
    slices (orienmask_yolo_fpnplus.py:78-86): the same products in the same order -- bit-identical."""
    from orienmask_amd.pack import conv_weights_split
    B, H, W, cout, segs = case
    L = omlib.load()
    g = torch.Generator().manual_seed(B * 1000 + H)
    cin = sum(c for c, _, _ in segs)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    ws, e = conv_weights_split(w, cout)
    wd = ws.to(dev)
    sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
    hp = (torch.randn(cout, generator=g) * 0.2).to(dev)
    bufs, views, parts = [], [], []
    for c, up, ps in segs:
        buf = torch.randn(B, H // up, W // up, ps, generator=g).to(dev)
        off = ps - c                                       # the slice sits at the END of its buffer's pixel
        bufs.append(buf)
        views.append(buf.view(-1)[off:])
        parts.append(buf[..., off:].repeat_interleave(up, 1).repeat_interleave(up, 2))
    cat = torch.cat(parts, -1).contiguous()
    assert cat.shape == (B, H, W, cin)
    want = torch.full((B, H, W, cout), float("nan"), device=dev)
    omlib.check(L.om_conv2d_split(_p(cat), B, H, W, cin, cin, _p(wd), _p(sp), _p(hp), cout, 1, 1, 1, None, 0, _p(want), cout, 0, 1,
                                  0, 0, None, omlib.current_stream_ptr(dev)), "om_conv2d_split")
    got = torch.full((B, H, W, cout), float("nan"), device=dev)
    n = len(segs)
    ptrs = (ctypes.c_void_p * n)(*[v.data_ptr() for v in views])
    ints = lambda vals: (ctypes.c_int * n)(*vals)
    omlib.check(L.om_conv2d_split_gather(n, ptrs, ints([c for c, _, _ in segs]), ints([ps for _, _, ps in segs]),
                                         ints([up for _, up, _ in segs]), B, H, W, _p(wd), _p(sp), _p(hp), cout, 1, _p(got), cout,
                                         None, omlib.current_stream_ptr(dev)), "om_conv2d_split_gather")
    torch.cuda.synchronize()
    assert torch.isfinite(want).all()
    assert torch.equal(got, want)
    # refusals: a segment that is not whole 32-channel chunks, an up factor that does not divide the map
    bad = L.om_conv2d_split_gather(1, ptrs, ints([48] + [32] * (n - 1)), ints([ps for _, _, ps in segs]), ints([1] * n), B, H, W,
                                   _p(wd), _p(sp), _p(hp), cout, 1, _p(got), cout, None, omlib.current_stream_ptr(dev))
    assert bad != 0


@pytest.mark.parametrize("prec", ["f32_split", "f16"])
def test_forward_gather_equals_replicated_concat(dev, prec):
    """Split-operand and fp16 forwards: the up-sampling-on-read form (routes and skips stored once at their own resolution, neck16.0 / neck8.0 /
    neck4.0 gathering them) against the replicated-concat form of the same library (set_upsample_on_read(False): what the other
    precisions run) -- all four heads bit-identical, bs=3 at 544x544 and a 64x96 image; keeping the activations
    (om_layer_output_view reports slices of the concat buffers) also selects the replicated form."""
    sd = synth.synth_state_dict(11, obj_bias=-16.0, head_gain=4.0)
    for shape, seed in (((3, 544, 544), 31), ((2, 64, 96), 32)):
        x = synth.synth_image_batch(seed, *shape).to(dev)
        net = _hip_model(sd, dev).set_precision(prec)
        kinds = dict(net.layer_kernels(*shape))
        assert "gather" in kinds["neck4.0"] and "gather" in kinds["neck8.0"] and "gather" in kinds["neck16.0"]
        with torch.no_grad():
            got = [(a.clone(), b.clone()) for a, b in net(x)]
            net.set_upsample_on_read(False)
            assert "gather" not in dict(net.layer_kernels(*shape))["neck4.0"]
            want = [(a.clone(), b.clone()) for a, b in net(x)]
            net.set_upsample_on_read(True).keep_activations(True)
            assert "gather" not in dict(net.layer_kernels(*shape))["neck4.0"]
        torch.cuda.synchronize()
        for (ga, gb), (wa, wb) in zip(got, want):
            assert torch.isfinite(wa).all() and torch.isfinite(wb).all()
            assert torch.equal(ga, wa) and torch.equal(gb, wb)


@pytest.mark.parametrize("mode", ["upsample", "nchw"])
def test_conv_split_output_modes(dev, mode):
    """The split-operand kernel's other two epilogues: nearest up-sampling into a channel slice of a concat buffer, and the
    NCHW orientation head (18 channels, no activation)."""
    from orienmask_amd.pack import conv_weights_split
    L = omlib.load()
    g = torch.Generator().manual_seed(77)
    B, H, W = 2, 6, 5
    cin, cout, up, leaky = (256, 64, 4, 1) if mode == "upsample" else (256, 18, 1, 0)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    want = torch.nn.functional.conv2d(x.double(), w.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    cpad = (cout + 31) // 32 * 32
    ws, e = conv_weights_split(w, cpad)
    sp = torch.zeros(cpad); sp[:cout] = scale
    sp = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float()
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, sd_, hd = ws.to(dev), sp.to(dev), hp.to(dev)
    if mode == "upsample":
        want = torch.nn.functional.interpolate(want, scale_factor=float(up), mode="nearest")
        buf = torch.full((B, H * up, W * up, 256), 7.0, device=dev)
        view = buf[..., 128:]
        rc = L.om_conv2d_split(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, 1, 1, leaky, None, 0,
                               ctypes.c_void_p(view.data_ptr()), 256, 1, up, 0, 0, None, omlib.current_stream_ptr(dev))
        omlib.check(rc, "om_conv2d_split")
        got = buf[..., 128:128 + cout].cpu().permute(0, 3, 1, 2).double()
        assert (torch.cat([buf[..., :128], buf[..., 128 + cout:]], -1) == 7.0).all()
    else:
        out = torch.full((B, cout, H, W), float("nan"), device=dev)
        rc = L.om_conv2d_split(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, 1, 1, leaky, None, 0,
                               _p(out), cout, 2, 1, 0, 0, None, omlib.current_stream_ptr(dev))
        omlib.check(rc, "om_conv2d_split")
        got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert _rel_err(got, want) < 2e-6


WINO_CASES = [
    # B, H, W, cin, cout, leaky, residual
    (2, 16, 16, 32, 64, 1, True),
    (1, 17, 17, 64, 128, 1, False),      # odd size: the last tile row/column is half outside
    (3, 8, 12, 128, 256, 1, True),
    (2, 34, 34, 128, 256, 0, False),
    (5, 5, 7, 256, 512, 1, True),
    (1, 68, 68, 32, 64, 1, True),
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_layer_matches_torch(dev, case):
    """Winograd F(2x2,3x3) path vs float64 direct convolution."""
    from orienmask_amd.pack import winograd_weights
    B, H, W, cin, cout, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, cout, H, W, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 63) // 64 * 64
    ud = winograd_weights(w, cpad).contiguous().to(dev)
    sp = torch.zeros(cpad); sp[:cout] = scale
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    sd_, hd = sp.to(dev), hp.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    scratch = torch.empty(L.om_conv2d_winograd_scratch_bytes(B, H, W, cin), dtype=torch.uint8, device=dev)
    rc = L.om_conv2d_winograd(_p(xd), B, H, W, cin, cin, _p(ud), _p(sd_), _p(hd), cout, leaky,
                              _p(rd) if use_res else None, cout if use_res else 0, _p(out), cout, _p(scratch),
                              scratch.numel(), omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_winograd")
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert _rel_err(got, want) < 5e-6, case


WINO24_CASES = WINO_CASES + [
    (2, 9, 13, 128, 64, 1, False),       # width and height not multiples of the 2 x 4 tile
    (1, 3, 5, 64, 128, 0, True),
    (4, 34, 34, 256, 512, 1, True),
    (8, 68, 68, 32, 512, 1, True),       # 584 tiles for 512 resident workgroups: the stream-K form (tiles cut between slots)
]


@pytest.mark.parametrize("case", WINO24_CASES)
def test_winograd24_layer_matches_torch(dev, case):
    """Winograd F(2x4,3x3) path vs float64 direct convolution (measured error 1.5e-6 of scale; F(2x2): 3.6e-7)."""
    from orienmask_amd.pack import winograd_weights
    B, H, W, cin, cout, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case) + 17)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, cout, H, W, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 63) // 64 * 64
    ud = winograd_weights(w, cpad, 24).contiguous().to(dev)
    sp = torch.zeros(cpad); sp[:cout] = scale
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    sd_, hd = sp.to(dev), hp.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    scratch = torch.empty(L.om_conv2d_winograd24_scratch_bytes(B, H, W, cin), dtype=torch.uint8, device=dev)
    rc = L.om_conv2d_winograd24(_p(xd), B, H, W, cin, cin, _p(ud), _p(sd_), _p(hd), cout, leaky,
                                _p(rd) if use_res else None, cout if use_res else 0, _p(out), cout, _p(scratch),
                                scratch.numel(), omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_winograd24")
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert _rel_err(got, want) < 1e-5, case


@pytest.mark.parametrize("case", WINO24_CASES)
def test_winograd24_split_layer_matches_torch(dev, case):
    """The same layers with SPLIT operands (hi/lo fp16 pairs, three fp16 MFMAs per product group, fp32 accumulate;
    include/orienmask_hip.h: om_model_set_precision) vs float64 direct convolution: within the fp32-operand kernel's own
    bound, and no more than 2x that kernel's error on the same inputs + 1e-6 (the hi/lo representation is ~2^-22)."""
    from orienmask_amd.pack import winograd_weights, winograd_weights_split
    B, H, W, cin, cout, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case) + 17)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, cout, H, W, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 63) // 64 * 64
    sp = torch.zeros(cpad); sp[:cout] = scale
    hp = torch.zeros(cpad); hp[:cout] = shift
    us, e = winograd_weights_split(w, cpad)
    sps = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    scratch = torch.empty(L.om_conv2d_winograd24_scratch_bytes(B, H, W, cin), dtype=torch.uint8, device=dev)
    errs = {}
    for mode in ("f32", "split"):
        out = torch.full((B, H, W, cout), float("nan"), device=dev)
        if mode == "f32":
            ud, sd_ = winograd_weights(w, cpad, 24).contiguous().to(dev), sp.to(dev)
            fn = L.om_conv2d_winograd24
        else:
            ud, sd_ = us.to(dev), sps.to(dev)
            fn = L.om_conv2d_winograd24_split
        hd = hp.to(dev)
        extra = (None,) if mode == "split" else ()          # status_dev
        rc = fn(_p(xd), B, H, W, cin, cin, _p(ud), _p(sd_), _p(hd), cout, leaky, _p(rd) if use_res else None,
                cout if use_res else 0, _p(out), cout, _p(scratch), scratch.numel(), *extra, omlib.current_stream_ptr(dev))
        omlib.check(rc, "om_conv2d_winograd24 " + mode)
        got = out.cpu().permute(0, 3, 1, 2).double()
        assert torch.isfinite(got).all(), mode
        errs[mode] = _rel_err(got, want)
    print("winograd24 %s: fp32 operands %.2e, split operands %.2e" % (case, errs["f32"], errs["split"]))
    assert errs["split"] < 1e-5, (case, errs)
    assert errs["split"] <= 2 * errs["f32"] + 1e-6, (case, errs)


WINO14_CASES = WINO24_CASES + [
    (1, 17, 17, 512, 128, 1, True),      # blocks spanning... one image, many channel chunks
    (4, 17, 17, 64, 64, 1, False),       # 5 tile columns, 25-row blocks across image boundaries (pad rows inside a block)
    (2, 34, 34, 32, 192, 1, True),       # 9 tile columns (36 > 34 pixels), three N tiles
    (1, 40, 136, 16, 64, 0, False),      # two column blocks of 17, one chunk
    (3, 7, 9, 48, 70, 1, True),          # cout not a multiple of 64 (scalar stores on the last quad), tiny image
]


@pytest.mark.parametrize("case", WINO14_CASES)
def test_wino14_split_layer_matches_torch(dev, case):
    """conv_wino14.hip -- the fused F(4,3)-along-the-rows form with split operands that om_forward runs for the stride-1 3x3
    layers in precision mode 1 (input transform inside the kernel, no transformed input in memory) -- vs float64 direct
    convolution: within the F(2x4) kernels' bound, and no worse than the fp32-operand F(2x4) kernel on the same inputs."""
    from orienmask_amd.pack import winograd14_weights_split
    B, H, W, cin, cout, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case) + 23)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, cout, H, W, generator=g) if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 63) // 64 * 64
    us, e = winograd14_weights_split(w, cpad)
    sp = torch.zeros(cpad); sp[:cout] = scale
    sps = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float().to(dev)
    hp = torch.zeros(cpad); hp[:cout] = shift
    hd = hp.to(dev)
    # the input as a channel slice of a wider buffer (concat views), the output with a pixel stride of its own
    xbuf = torch.full((B, H, W, cin + 16), 9.0, device=dev)
    xbuf[..., 8:8 + cin] = x.permute(0, 2, 3, 1).to(dev)
    xv = xbuf[..., 8:]
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    ostride = cout + (4 - cout % 4) % 4 + 4
    out = torch.full((B, H, W, ostride), float("nan"), device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ud = us.to(dev)
    rc = L.om_conv2d_wino14_split(ctypes.c_void_p(xv.data_ptr()), B, H, W, cin, cin + 16, _p(ud), _p(sps), _p(hd), cout, leaky,
                                  _p(rd) if use_res else None, cout if use_res else 0, _p(out), ostride, _p(status),
                                  omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_wino14_split")
    got = out[..., :cout].cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all() and int(status.item()) == 0
    assert torch.isnan(out[..., cout:]).all()                       # nothing written beyond the layer's channels
    err = _rel_err(got, want)
    print("wino14 split %s: %.2e" % (case, err))
    assert err < 5e-6, (case, err)


WINO14_DUAL_CASES = [
    # B, H, W, cin, cout, leaky, residual -- all with an even number >= 2 of 16-channel chunks (the dual-role kernel's domain)
    (2, 34, 34, 32, 192, 1, True),       # two chunks: every chunk is a tile's first or last; three N tiles
    (4, 17, 17, 64, 64, 1, False),       # 25-row blocks across image boundaries, four chunks
    (1, 40, 136, 32, 64, 0, False),      # four column blocks of 9
    (3, 68, 68, 128, 256, 1, True),      # the network's 68 x 68 layers: more tiles than one round of 256 workgroups
    (2, 17, 17, 512, 128, 1, True),      # 32 chunks
    (1, 7, 9, 32, 70, 1, False),         # tiny image: (R + 2) Ct below 128 entries, cout not a multiple of 64
    (9, 136, 136, 64, 128, 1, False),    # many tiles per workgroup: the chunk stream across tiles, the two-ahead ticket
]


@pytest.mark.parametrize("case", WINO14_DUAL_CASES)
def test_wino14_dual_equals_twelve_wave(dev, case):
    """conv_wino14d.hip (round 5: four dual-role waves, accumulators owned by name) against conv_wino14.hip (eight consumer + four
    producer waves) on the same inputs: the same sequence of fp32 operations per output, so bit-identical outputs -- and both within
    the fused form's bound of the float64 convolution (/root/reference/model/base.py:104-137)."""
    from orienmask_amd.pack import winograd14_weights_split
    B, H, W, cin, cout, leaky, use_res = case
    L = omlib.load()
    if not L.om_wino14_dual_built():
        assert L.om_set_wino14_variant(1) != 0          # loud, not silently the other kernel
        pytest.skip("the default library does not contain the dual-role kernel (run the session with W14D=1 in the environment)")
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, H, W, cout, generator=g) if use_res else None
    cpad = (cout + 63) // 64 * 64
    us, e = winograd14_weights_split(w, cpad)
    sp = torch.zeros(cpad); sp[:cout] = scale
    sps = (sp.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float().to(dev)
    hp = torch.zeros(cpad); hp[:cout] = shift
    hd, ud, xd = hp.to(dev), us.to(dev), x.to(dev)
    rd = res.to(dev) if use_res else None
    ostride = cout + (4 - cout % 4) % 4
    outs = []
    try:
        for variant in (0, 1):
            omlib.check(L.om_set_wino14_variant(variant), "om_set_wino14_variant")
            for rep in range(2 if variant else 1):           # the dual kernel twice: nothing depends on what the LDS held
                out = torch.full((B, H, W, ostride), float("nan"), device=dev)
                status = torch.zeros(1, dtype=torch.int32, device=dev)
                rc = L.om_conv2d_wino14_split(_p(xd), B, H, W, cin, cin, _p(ud), _p(sps), _p(hd), cout, leaky,
                                              _p(rd) if use_res else None, cout if use_res else 0, _p(out), ostride, _p(status),
                                              omlib.current_stream_ptr(dev))
                omlib.check(rc, "om_conv2d_wino14_split")
                torch.cuda.synchronize()
                assert int(status.item()) == 0
                outs.append(out.cpu())
    finally:
        L.om_set_wino14_variant(0)
    ref = outs[0]
    assert torch.isfinite(ref[..., :cout]).all()
    for o in outs[1:]:
        assert torch.equal(o[..., :cout], ref[..., :cout]), (case, (o[..., :cout] - ref[..., :cout]).abs().max())
        assert torch.isnan(o[..., cout:]).all()
    want = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.permute(0, 3, 1, 2).double()
    err = _rel_err(outs[1][..., :cout].permute(0, 3, 1, 2).double(), want)
    assert err < 5e-6, (case, err)


WINO14_WIDE_CASES = [
    # B, H, W, cin, cout, leaky, residual
    (3, 17, 17, 512, 1024, 1, True),     # the layers om_forward runs this way (conv6.*.conv.1: residual; blocks span images)
    (2, 17, 17, 512, 1024, 1, False),    # neck32.1 / neck32.3 / bbox_head32.0
    (2, 10, 13, 64, 128, 1, True),       # W not a multiple of 4, one pair of N tiles, four chunks
    (1, 34, 34, 32, 256, 0, False),      # 9 tile columns, two chunks, no activation
    (5, 6, 7, 16, 384, 1, True),         # ONE chunk (nothing to prefetch), three pairs of N tiles, tiny images
    (2, 40, 72, 48, 128, 1, False),      # two column blocks per row (18 tile columns), an odd number of chunks
]


@pytest.mark.parametrize("case", WINO14_WIDE_CASES)
def test_wino14_wide_equals_fused(dev, case):
    """The two-kernel wide form of the stride-1 3x3 layer (round 6, conv_wino14.hip: wino14_v_kernel writes the transformed input,
    wino14_wide_kernel multiplies 128 x 128 tiles reading it by LDS-DMA) against the fused kernel on the same inputs: the same
    products in the same order, so BIT-IDENTICAL outputs -- twice in a row (nothing depends on what LDS or the scratch held), with
    the scratch poisoned in between; the fused kernel itself is held to the float64 convolution by
    test_wino14_split_layer_matches_torch (/root/reference/model/base.py:104-137)."""
    from orienmask_amd.pack import winograd14_weights_split
    B, H, W, cin, cout, leaky, use_res = case
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(case) + 17)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    res = torch.randn(B, H, W, cout, generator=g) if use_res else None
    us, e = winograd14_weights_split(w, cout)
    sps = (scale.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double()[:cout])).float().to(dev)
    hd, ud, xd = shift.to(dev), us.to(dev), x.to(dev)
    rd = res.to(dev) if use_res else None
    nbytes = L.om_conv2d_wino14_wide_scratch_bytes(B, H, W, cin)
    assert nbytes >= (cin // 16) * 6 * B * (H + 2) * ((W + 3) // 4) * 64
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = omlib.current_stream_ptr(dev)

    def run(wide):
        out = torch.full((B, H, W, cout), float("nan"), device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        if wide:
            rc = L.om_conv2d_wino14_wide(_p(xd), B, H, W, cin, cin, _p(ud), _p(sps), _p(hd), cout, leaky, _p(rd) if use_res else None,
                                         cout if use_res else 0, _p(out), cout, _p(scratch), nbytes, _p(status), st)
        else:
            rc = L.om_conv2d_wino14_split(_p(xd), B, H, W, cin, cin, _p(ud), _p(sps), _p(hd), cout, leaky, _p(rd) if use_res else None,
                                          cout if use_res else 0, _p(out), cout, _p(status), st)
        omlib.check(rc, "wino14 wide" if wide else "wino14 fused")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        return out.cpu()

    ref = run(False)
    assert torch.isfinite(ref).all()
    scratch.fill_(0xFF)                  # NaN halves wherever the pre-pass does not write and the consumer still reads
    a = run(True)
    scratch.fill_(0x7C)                  # +inf halves
    b = run(True)
    assert torch.equal(a, ref), (case, (a - ref).abs().max())
    assert torch.equal(b, ref), case
    # what it refuses, it refuses loudly: a single 64-channel N tile, a scratch that is too small
    out = torch.empty(B, H, W, 64, device=dev)
    assert L.om_conv2d_wino14_wide(_p(xd), B, H, W, cin, cin, _p(ud), _p(sps), _p(hd), 64, leaky, None, 0, _p(out), 64, _p(scratch), nbytes,
                                   None, st) != 0
    assert L.om_conv2d_wino14_wide(_p(xd), B, H, W, cin, cin, _p(ud), _p(sps), _p(hd), cout, leaky, None, 0, _p(ref.to(dev)), cout,
                                   _p(scratch), nbytes - 256, None, st) != 0


def test_forward_wide_3x3_switch_is_bit_identical(dev):
    """om_forward runs the stride-1 3x3 layers with at least 512 input channels (the 1/32-scale ones) in the two-kernel wide form
    (om_set_wino14_wide, default on): the layer table says so, the workspace grows by their transformed input, and all six head
    tensors are bit-identical to a forward with the switch off; the previous setting is restored."""
    L = omlib.load()
    sd = synth.synth_state_dict(13, obj_bias=-16.0, head_gain=4.0)
    # 140 strips of 32 x 544: 140 x 3 padded rows of 5 tile columns at 1/32 scale = 17 row blocks x 16 N tiles = 272 tiles of the fused
    # kernel -- more than the 256 CUs, which is where om_forward switches (a batch of 3 such images stays with the fused kernel)
    x = synth.synth_image_batch(34, 140, 32, 544).to(dev)
    net = _hip_model(sd, dev, "f32_split")
    was = L.om_get_wino14_wide()
    try:
        omlib.check(L.om_set_wino14_wide(1), "om_set_wino14_wide")
        assert not any(v.startswith("wino14_wide_kernel") for _, v in net.layer_kernels(3, 32, 544))
        k_on = dict(net.layer_kernels(140, 32, 544))
        with torch.no_grad():
            pred = net(x)
            on = [(b.clone(), o.clone()) for b, o in pred]
        assert pred.flags() == 0
        omlib.check(L.om_set_wino14_wide(0), "om_set_wino14_wide")
        net._workspace.clear(); net._slot_workspaces.clear()          # the layout differs (no V scratch)
        k_off = dict(net.layer_kernels(140, 32, 544))
        with torch.no_grad():
            off = net(x)
        for (a, b_), (c, d) in zip(on, off):
            assert torch.equal(a, c) and torch.equal(b_, d)
    finally:
        L.om_set_wino14_wide(was)
        net._workspace.clear(); net._slot_workspaces.clear()
    wide = sorted(k for k, v in k_on.items() if v.startswith("wino14_wide_kernel"))
    assert wide == sorted(["backbone.conv6.%d.conv.1" % i for i in range(1, 5)] + ["neck32.1", "neck32.3", "bbox_head32.0"]), wide
    assert all(k_off[k].startswith("wino14_split_kernel") for k in wide)
    assert k_on["backbone.conv5.3.conv.1"].startswith("wino14_split_kernel")       # 256 input channels: the fused kernel


@pytest.mark.parametrize("gain,finite", [(100.0, True), (3.0e4, False)])
def test_split_operand_range(dev, gain, finite):
    """The documented range of split operands (include/orienmask_hip.h: om_model_set_precision): activations 100x larger than a
    BatchNorm-ed network produces are still exact to fp32 level (the error is relative, so it does not grow with the scale),
    and inputs whose transform exceeds fp16's 65504 give non-finite outputs AND raise OM_STATUS_SPLIT_RANGE in the status word
    -- loud, never a silently wrong number."""
    from orienmask_amd.pack import winograd_weights_split
    B, H, W, cin, cout = 2, 16, 20, 64, 64
    L = omlib.load()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, H, W, generator=g) * gain
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    us, e = winograd_weights_split(w, cout)
    sps = torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double()).float().to(dev)
    hd = torch.zeros(cout, device=dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.full((B, H, W, cout), float("nan"), device=dev)
    scratch = torch.empty(L.om_conv2d_winograd24_scratch_bytes(B, H, W, cin), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = L.om_conv2d_winograd24_split(_p(xd), B, H, W, cin, cin, _p(us.to(dev)), _p(sps), _p(hd), cout, 0, None, 0, _p(out), cout,
                                      _p(scratch), scratch.numel(), _p(status), omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_winograd24_split")
    got = out.cpu().permute(0, 3, 1, 2).double()
    if finite:
        assert torch.isfinite(got).all()
        assert _rel_err(got, want) < 5e-6
        assert int(status.item()) == 0
    else:
        assert not torch.isfinite(got).all()
        assert int(status.item()) == omlib.OM_STATUS_SPLIT_RANGE
    # the 1x1 / stride-2 kernel: raw activations beyond 65504 (no transform in front of them)
    from orienmask_amd.pack import conv_weights_split
    w1 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    ws, e1 = conv_weights_split(w1, cout)
    sp1 = torch.pow(torch.tensor(2.0, dtype=torch.float64), -e1.double()).float().to(dev)
    x1 = (x / gain * (100.0 if finite else 7.0e4)).permute(0, 2, 3, 1).contiguous().to(dev)
    status.zero_()
    out1 = torch.empty((B, H, W, cout), device=dev)
    rc = L.om_conv2d_split(_p(x1), B, H, W, cin, cin, _p(ws.to(dev)), _p(sp1), _p(hd), cout, 1, 1, 0, None, 0, _p(out1), cout, 0, 1,
                           0, 0, _p(status), omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_split")
    assert bool(torch.isfinite(out1).all()) == finite
    assert int(status.item()) == (0 if finite else omlib.OM_STATUS_SPLIT_RANGE)


def test_stem_matches_torch(dev):
    L = omlib.load()
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 40, 72
    x = torch.rand(B, 3, H, W, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.2
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.1)
    wp = w.permute(0, 2, 3, 1).reshape(32, 27).contiguous().to(dev)
    out = torch.full((B, H, W, 32), float("nan"), device=dev)
    xd, scd, shd = x.to(dev), scale.to(dev), shift.to(dev)      # keep the device tensors alive over the call
    rc = L.om_conv2d_stem(_p(xd), B, H, W, _p(wp), _p(scd), _p(shd), 32, _p(out), omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_stem")
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert _rel_err(got, want) < 2e-6


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 34, 50), (3, 16, 32), (1, 96, 160)])
def test_stem2_split_matches_two_kernels(dev, shape):
    """conv_stem2.hip -- backbone.conv1 + backbone.conv2.0 (3x3 stride 2, 32 -> 64) as ONE kernel in split-operand mode, conv1's
    activation never in memory, conv1 itself on the matrix pipe with split operands -- against float64 (the bound of the
    two-kernel path, 2e-6 of scale) and against om_conv2d_stem followed by om_conv2d_split (fp32 conv1 on the vector ALUs: same
    values to 2e-6), on sizes with partial tiles in both directions and an output view with a pixel stride of its own."""
    from orienmask_amd.pack import conv_weights_split
    B, H, W = shape
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 0.5
    w1 = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    sc1 = torch.rand(32, generator=g) + 0.5
    sh1 = torch.randn(32, generator=g) * 0.2
    w2 = torch.randn(64, 32, 3, 3, generator=g) / (32 * 9) ** 0.5
    sc2 = torch.rand(64, generator=g) + 0.5
    sh2 = torch.randn(64, generator=g) * 0.2
    a = torch.nn.functional.conv2d(x.double(), w1.double(), None, 1, 1) * sc1.double().view(1, -1, 1, 1) + sh1.double().view(1, -1, 1, 1)
    a = torch.where(a > 0, a, a * 0.1)
    want = torch.nn.functional.conv2d(a, w2.double(), None, 2, 1) * sc2.double().view(1, -1, 1, 1) + sh2.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.1)
    w1p = w1.permute(0, 2, 3, 1).reshape(32, 27).contiguous().to(dev)
    ws, e = conv_weights_split(w2, 64)
    sp2 = (sc2.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double())).float().to(dev)
    xd, sc1d, sh1d, wsd, sh2d = x.to(dev), sc1.to(dev), sh1.to(dev), ws.to(dev), sh2.to(dev)
    st = omlib.current_stream_ptr(dev)
    Ho, Wo = H // 2, W // 2
    # two kernels
    mid = torch.empty(B, H, W, 32, device=dev)
    omlib.check(L.om_conv2d_stem(_p(xd), B, H, W, _p(w1p), _p(sc1d), _p(sh1d), 32, _p(mid), st), "om_conv2d_stem")
    two = torch.full((B, Ho, Wo, 64), float("nan"), device=dev)
    omlib.check(L.om_conv2d_split(_p(mid), B, H, W, 32, 32, _p(wsd), _p(sp2), _p(sh2d), 64, 3, 2, 1, None, 0, _p(two), 64, 0, 1, 0, 0,
                                  None, st), "om_conv2d_split")
    # one kernel, into a channel slice of a wider buffer
    ostride = 80
    buf = torch.full((B, Ho, Wo, ostride), float("nan"), device=dev)
    view = buf[..., 8:]
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    omlib.check(L.om_conv2d_stem2_split(_p(xd), B, H, W, _p(w1p), _p(sc1d), _p(sh1d), _p(wsd), _p(sp2), _p(sh2d), 64, 1,
                                        ctypes.c_void_p(view.data_ptr()), ostride, _p(status), st), "om_conv2d_stem2_split")
    one = buf[..., 8:72]
    assert int(status.item()) == 0
    assert torch.isnan(buf[..., :8]).all() and torch.isnan(buf[..., 72:]).all()      # nothing outside the 64 channels
    e1 = _rel_err(one.cpu().permute(0, 3, 1, 2).double(), want)
    e2 = _rel_err(two.cpu().permute(0, 3, 1, 2).double(), want)
    e12 = _rel_err(one.cpu().double(), two.cpu().double())
    print("stem2 %s: one kernel %.2e, two kernels %.2e of scale against float64; against each other %.2e" % (shape, e1, e2, e12))
    assert e1 < 2e-6 and e1 <= 2 * e2 + 5e-7 and e12 < 2e-6


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 34, 50), (1, 96, 160)])
def test_stem3_split_equals_stem2_then_1x1(dev, shape):
    """conv_stem2_split_kernel with its optional third layer (round 5): backbone.conv2.1.conv.0 -- the 64 -> 32 1x1 convolution
    behind conv2.0 -- computed on each tile's outputs inside the same launch.  conv2.0's tensor must be the two-layer kernel's bit for
    bit, and the third layer's what om_conv2d_split computes from it, bit for bit (same split, same three products per 16 channels)."""
    from orienmask_amd.pack import conv_weights_split
    B, H, W = shape
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(shape) + 9)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 0.5
    w1 = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    sc1, sh1 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    w2 = torch.randn(64, 32, 3, 3, generator=g) / (32 * 9) ** 0.5
    sc2, sh2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    w3 = torch.randn(32, 64, 1, 1, generator=g) / 8.0
    sc3, sh3 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    w1p = w1.permute(0, 2, 3, 1).reshape(32, 27).contiguous().to(dev)
    ws2, e2 = conv_weights_split(w2, 64)
    ws3, e3 = conv_weights_split(w3, 32)
    sp2 = (sc2.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e2.double())).float().to(dev)
    sp3 = (sc3.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e3.double())).float().to(dev)
    xd, sc1d, sh1d, ws2d, sh2d, ws3d, sh3d = x.to(dev), sc1.to(dev), sh1.to(dev), ws2.to(dev), sh2.to(dev), ws3.to(dev), sh3.to(dev)
    st = omlib.current_stream_ptr(dev)
    Ho, Wo = H // 2, W // 2
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    two = torch.full((B, Ho, Wo, 64), float("nan"), device=dev)
    omlib.check(L.om_conv2d_stem2_split(_p(xd), B, H, W, _p(w1p), _p(sc1d), _p(sh1d), _p(ws2d), _p(sp2), _p(sh2d), 64, 1,
                                        _p(two), 64, _p(status), st), "om_conv2d_stem2_split")
    sep = torch.full((B, Ho, Wo, 32), float("nan"), device=dev)
    omlib.check(L.om_conv2d_split(_p(two), B, Ho, Wo, 64, 64, _p(ws3d), _p(sp3), _p(sh3d), 32, 1, 1, 1, None, 0, _p(sep), 32, 0, 1, 0, 0,
                                  _p(status), st), "om_conv2d_split")
    out2 = torch.full((B, Ho, Wo, 64), float("nan"), device=dev)
    buf3 = torch.full((B, Ho, Wo, 48), float("nan"), device=dev)          # the third layer's view: channels 8..39 of a wider buffer
    omlib.check(L.om_conv2d_stem3_split(_p(xd), B, H, W, _p(w1p), _p(sc1d), _p(sh1d), _p(ws2d), _p(sp2), _p(sh2d), 64, 1, _p(out2), 64,
                                        _p(ws3d), _p(sp3), _p(sh3d), 32, 1, ctypes.c_void_p(buf3[..., 8:].data_ptr()), 48, _p(status), st),
                "om_conv2d_stem3_split")
    assert int(status.item()) == 0
    assert torch.equal(out2, two)
    assert torch.isnan(buf3[..., :8]).all() and torch.isnan(buf3[..., 40:]).all()
    assert torch.equal(buf3[..., 8:40], sep)
    a = torch.nn.functional.conv2d(x.double(), w1.double(), None, 1, 1) * sc1.double().view(1, -1, 1, 1) + sh1.double().view(1, -1, 1, 1)
    a = torch.where(a > 0, a, a * 0.1)
    b2 = torch.nn.functional.conv2d(a, w2.double(), None, 2, 1) * sc2.double().view(1, -1, 1, 1) + sh2.double().view(1, -1, 1, 1)
    b2 = torch.where(b2 > 0, b2, b2 * 0.1)
    want = torch.nn.functional.conv2d(b2, w3.double()) * sc3.double().view(1, -1, 1, 1) + sh3.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.1)
    assert _rel_err(buf3[..., 8:40].cpu().permute(0, 3, 1, 2).double(), want) < 4e-6


# ------------------------------------------------------------------------------------------------
# forward
# ------------------------------------------------------------------------------------------------
def _hip_model(sd, dev, precision="f32"):
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision(precision)
    net.load_state_dict(sd, strict=True)
    return net.to(dev)


def _check_forward_fixture(out, g, fname, precision, dev, size, batch):
    """Head tensors against the digests / tensors the reference produced, then the composed path against its detections."""
    got = dict(bbox32=out[0][0], bbox16=out[1][0], bbox8=out[2][0],
               oriens=torch.cat([out[0][1], out[1][1], out[2][1]], 1))
    for k, t in got.items():
        assert list(t.shape) == g[k + "_shape"].tolist(), k
        flat = t.contiguous().cpu().reshape(-1)
        scale = float(g[k + "_absmax"])
        samp = flat[torch.from_numpy(g[k + "_idx"])].numpy()
        assert np.max(np.abs(samp - g[k + "_samples"])) <= REL_TOL * scale, k
        assert abs(flat.double().sum().item() - g[k + "_sum"][0]) <= REL_TOL * g[k + "_sum"][1], k
        if k in g.files:
            assert np.max(np.abs(flat.numpy() - g[k].reshape(-1))) <= REL_TOL * scale, k
    # the composed path: HIP forward -> HIP postprocess against the reference's END-TO-END detections of the same image
    # (its own forward feeding its own postprocess)
    res = _hip_post(size, dev)(out)
    assert len(res) == batch
    pc = post_cfg(size)
    oracle_post = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                                      conf_thresh=pc["conf_thresh"])
    heads_cpu = [(b_.cpu(), o_.cpu()) for b_, o_ in out]
    for b, r in enumerate(res):
        _check_detections_composed(r, g["bbox_det%d" % b], g["cls_det%d" % b],
                                   unpack_masks(g["mask%d" % b], g["maskshape%d" % b]), (fname, precision, b),
                                   margin_ctx=_margin_ctx(oracle_post, heads_cpu, b))


@pytest.mark.parametrize("precision", ["f32", "f32_split"])
@pytest.mark.parametrize("fname", golden_files("fwd_"))
def test_forward_matches_reference_golden(dev, fname, precision):
    """HIP forward vs tensors the real reference produced (tests/golden/fwd_*.npz), with fp32 operands and with split operands
    (every convolution but the stem; the stride-1 3x3 layers run F(2x4) at every batch size in that mode)."""
    g = np.load(os.path.join(GOLDEN, fname))
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    sd, x = fixture_weights_and_input(g)       # fwd_stress_*: heavy-tailed BatchNorm scales, 1e-20 / zero rows, saturated input
    net = _hip_model(sd, dev, precision)
    with torch.no_grad():
        out = net(x.to(dev))
    torch.cuda.synchronize()
    assert out.flags() == 0, "a fixture must not need the fp32-operand fallback (it would test nothing of the split kernels)"
    _check_forward_fixture(out, g, fname, precision, dev, size, batch)


_ORACLE_CACHE = {}


def _oracle_once(key, fn):
    """The CPU oracle's tensors for a seeded case, computed once per session (the same case runs in several precisions)."""
    if key not in _ORACLE_CACHE:
        with torch.no_grad():
            _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("precision", ["f32", "f32_split"])
def test_forward_matches_oracle_and_layouts(dev, precision):
    """Full head tensors vs the oracle on a non-square input, with fp32 operands and in the plugin's default precision; also pins
    the returned layouts."""
    sd = synth.synth_state_dict(5, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(6, 3, 128, 192)
    net = _hip_model(sd, dev, precision)
    with torch.no_grad():
        out = net(x.to(dev))
    assert out.flags() == 0
    ref = _oracle_once(("fwd", 5, 6, 3, 128, 192), lambda: R.forward(sd, x))
    for (gb, go), (rb, ro) in zip(out, ref):
        assert gb.shape == rb.shape and go.shape == ro.shape
        assert gb.stride(1) == 1                      # channels-last box heads
        assert go.stride() == (18 * go.shape[2] * go.shape[3], go.shape[2] * go.shape[3], go.shape[3], 1)
        assert _rel_err(gb.cpu(), rb) < REL_TOL
        assert _rel_err(go.cpu(), ro) < REL_TOL


def test_yolo_variant_matches_reference_golden(dev):
    """The non-Plus OrienMaskYOLO graph vs tensors the reference's own model produced."""
    from orienmask_amd.model import OrienMaskYOLO
    g = np.load(os.path.join(GOLDEN, "yolo_fwd.npz"))
    for name in ("y96_b2", "y128x160_b1"):
        wseed, xseed, batch, h, w = (int(v) for v in g[name + "_meta"])
        sd = synth.synth_state_dict(wseed, obj_bias=-16.0, head_gain=4.0, model="OrienMaskYOLO")
        net = OrienMaskYOLO(3, 80).eval()
        net.load_state_dict(sd, strict=True)
        net = net.to(dev)
        with torch.no_grad():
            out = net(synth.synth_image_batch(xseed, batch, h, w).to(dev))
        got = dict(bbox32=out[0][0], bbox16=out[1][0], bbox8=out[2][0],
                   oriens=torch.cat([out[0][1], out[1][1], out[2][1]], 1))
        for k, t in got.items():
            want = torch.from_numpy(g["%s_%s" % (name, k)])
            assert t.shape == want.shape
            assert _rel_err(t.cpu(), want) < REL_TOL, (name, k)


def test_forward_is_batch_invariant(dev):
    """Size-independent property at the full 544x544 size: an image's outputs do not depend on what else is in the batch
    (bit-exact) AS LONG AS both batches are on the same side of the Winograd switch (om_forward runs the stride-1 3x3 layers
    with F(2x2,3x3) below 1700 1/32-scale cells, i.e. bs < 6 at 544x544, and F(2x4,3x3) from there on: across the switch the
    results agree to ~1e-6 of scale, not to the bit -- test_forward_across_the_winograd_switch); repeated runs are
    bit-identical."""
    sd = synth.synth_state_dict(7, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(8, 4, 544, 544).to(dev)
    net = _hip_model(sd, dev)
    with torch.no_grad():
        full = net(x)
        again = net(x)
        single = net(x[2:3])
    for (fb, fo), (ab, ao), (sb, so) in zip(full, again, single):
        assert torch.equal(fb, ab) and torch.equal(fo, ao)
        assert torch.equal(fb[2:3], sb) and torch.equal(fo[2:3], so)
        assert torch.isfinite(fb).all() and torch.isfinite(fo).all()


def test_forward_split_is_batch_invariant(dev):
    """With split operands every size runs the same kernels (the fused F(4,3) form at any batch -- whose blocks of padded rows
    fall differently across the images of another batch -- and one summation order per output element in the implicit GEMM
    whatever the tile shape), so an image's head tensors are bit-identical alone, in a batch of 3 and in a batch of 7 -- across
    the size where precision 'f32' switches from F(2x2) to F(2x4) (and is only equal to ~1e-6)."""
    sd = synth.synth_state_dict(9, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(31, 7, 544, 544).to(dev)
    net = _hip_model(sd, dev, "f32_split")
    assert dict(net.layer_kernels(1, 544, 544))["orien_head.2"].startswith("wino14_split_kernel")
    with torch.no_grad():
        full = [(b.clone(), o.clone()) for b, o in net(x)]
        for idx in ([6], [2, 6, 4]):
            part = net(x[idx])
            for (fb, fo), (pb, po) in zip(full, part):
                assert torch.equal(fb[idx], pb) and torch.equal(fo[idx], po), idx


@pytest.mark.parametrize("precision,which", [("f32_split", 0), ("f16", 1)])
def test_stem_fusion_switch_is_bit_identical(dev, precision, which):
    """om_set_stem_fusion (round 6; before: environment variables read once): with the fusion of the first layers off -- the third
    layer outside the split-operand first-two-layers kernel, or the fp16 configuration's first two layers as separate launches --
    om_forward runs the kernels that were the only path before the fusions existed: the same bits in split-operand mode, the fp16
    configuration's own tolerance there (conv1's fp32 sums in another order before their one rounding); the layer table reports the
    kernel that runs; the previous setting is restored."""
    L = omlib.load()
    sd = synth.synth_state_dict(12, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(33, 2, 160, 224).to(dev)
    net = _hip_model(sd, dev, precision)
    was = L.om_get_stem_fusion(which)
    assert was in (0, 1) and L.om_get_stem_fusion(2) == -1 and L.om_set_stem_fusion(2, 1) != 0
    name = "backbone.conv2.1.conv.0" if which == 0 else "backbone.conv2.0"
    try:
        omlib.check(L.om_set_stem_fusion(which, 1), "om_set_stem_fusion")
        with torch.no_grad():
            fused = [(b.clone(), o.clone()) for b, o in net(x)]
        k_on = dict(net.layer_kernels(2, 160, 224))[name]
        omlib.check(L.om_set_stem_fusion(which, 0), "om_set_stem_fusion")
        assert L.om_get_stem_fusion(which) == 0
        with torch.no_grad():
            plain = net(x)
        k_off = dict(net.layer_kernels(2, 160, 224))[name]
        for (fb, fo), (pb, po) in zip(fused, plain):
            if which == 0:      # the third layer inside or behind the kernel: the same arithmetic, the same bits
                assert torch.equal(fb, pb) and torch.equal(fo, po)
            else:               # fp16 configuration: the fused kernel sums conv1 on the matrix pipe, the separate one on the vector
                                # ALUs -- fp32 sums in another order, each rounded ONCE to fp16 (test_stem2_f16_matches_two_kernels:
                                # > 97 % of conv2.0's outputs identical, the rest one fp16 step apart): the configuration's own bar
                assert _rel_err(fb.cpu(), pb.cpu()) < 5e-3 and _rel_err(fo.cpu(), po.cpu()) < 5e-3
        assert k_on == "(in the previous layer's kernel)" and k_off.startswith("conv_igemm"), (k_on, k_off)
    finally:
        L.om_set_stem_fusion(which, was)


def test_forward_across_the_winograd_switch(dev):
    """The same image in a batch of 2 (F(2x2,3x3)) and in a batch of 7 (F(2x4,3x3)): not bit-identical, but far inside the
    parity budget, and the composed detections agree in every index."""
    sd = synth.synth_state_dict(7, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(9, 7, 544, 544).to(dev)
    net = _hip_model(sd, dev)
    post = _hip_post((544, 544), dev)
    with torch.no_grad():
        big = net(x)
        det_big = {k: v.clone() for k, v in post(big)[1].items()}
        big = [(a[1:2].clone(), b[1:2].clone()) for a, b in big]
        small = net(x[:2])
        det_small = post(small)[1]
    assert dict(net.layer_kernels(7, 544, 544))["orien_head.2"].startswith("wino24")
    assert dict(net.layer_kernels(2, 544, 544))["orien_head.2"].startswith("wino_")
    for (bb, bo), (sb, so) in zip(big, small):
        assert _rel_err(bb, sb[1:2]) < 2e-5 and _rel_err(bo, so[1:2]) < 2e-5
    assert torch.equal(det_big["cls"], det_small["cls"])
    assert (det_big["bbox"] - det_small["bbox"]).abs().max().item() <= 1e-4


def test_forward_rejects_cpu_and_training(dev):
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64, device=dev))          # training mode
    net.eval()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))                      # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------------
# postprocess
# ------------------------------------------------------------------------------------------------
def _hip_post(size, dev, **kw):
    from orienmask_amd.eval import OrienMaskYOLOPostProcess
    cfg = post_cfg(size)
    cfg.update(kw)
    return OrienMaskYOLOPostProcess(device=dev, **cfg)


def _mask_iou(a, b):
    inter = np.logical_and(a, b).sum(); union = np.logical_or(a, b).sum()
    return 1.0 if union == 0 else inter / union


def _to_model_layout(heads, dev):
    """CPU NCHW heads -> the HIP model's own layouts (NHWC stride-256 boxes, one oriens buffer)."""
    out = []
    B = heads[0][0].shape[0]
    oriens = torch.cat([h[1] for h in heads], 1).contiguous().to(dev)
    os_ = torch.split(oriens, heads[0][1].shape[1], dim=1)
    for i, (b, _) in enumerate(heads):
        nh, nw = b.shape[2], b.shape[3]
        buf = torch.zeros(B, nh, nw, 256, device=dev)
        buf[..., :b.shape[1]] = b.permute(0, 2, 3, 1).to(dev)
        out.append((buf[..., :b.shape[1]].permute(0, 3, 1, 2), os_[i]))
    return tuple(out)


@pytest.mark.parametrize("layout", ["model", "plain"])
@pytest.mark.parametrize("fname", golden_files("post_"))
def test_postprocess_matches_reference_golden(dev, fname, layout):
    g = np.load(os.path.join(GOLDEN, fname))
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    pc = post_cfg(size)
    heads = synth.synth_heads(int(g["seed"]), batch, pc["grid_size"], regime=str(g["regime"]))
    dheads = _to_model_layout(heads, dev) if layout == "model" else tuple((b.to(dev), o.to(dev)) for b, o in heads)
    post = _hip_post(size, dev)
    res = post(dheads)
    torch.cuda.synchronize()
    assert len(res) == batch
    for b, r in enumerate(res):
        _check_detections(r, g["bbox%d" % b], g["cls%d" % b], unpack_masks(g["mask%d" % b], g["maskshape%d" % b]),
                          (fname, layout, b), exact_decode=True)


@pytest.mark.parametrize("regime,seed", [("mixed", 101), ("clustered", 102), ("sparse", 103), ("sparse_many", 104),
                                         ("dense", 105), ("empty", 106)])
def test_postprocess_matches_oracle_indices(dev, regime, seed):
    """Fresh seeds (not in the fixtures): same heads in, same candidate indices / classes out."""
    size = (160, 192)
    pc = post_cfg(size)
    heads = synth.synth_heads(seed, 3, pc["grid_size"], regime=regime)
    oracle = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                                 conf_thresh=pc["conf_thresh"])
    want = oracle(heads)
    post = _hip_post(size, dev)
    got = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
    for b, (r, w) in enumerate(zip(got, want)):
        assert r["bbox"].shape[0] == w["bbox"].shape[0], (regime, b)
        assert torch.equal(r["cls"].cpu(), w["cls"])
        assert torch.equal(post.last_keep[b].cpu().long(), w["keep"])
        _check_detections(r, w["bbox"].numpy(), w["cls"].numpy(), w["mask"].numpy(), (regime, b), exact_decode=True)


def test_ref_math_bit_exact(dev):
    """csrc/ref_math.h against torch-CPU at the reference's own call sites (postprocess.py:127-136): a strided view goes
    through torch's scalar loop (glibc expf), rows of C contiguous class logits through the vectorised loop (Sleef expf_u10)
    with a scalar tail of C mod 32 elements.  Bit-exact on 2 M values each, incl. the saturation ends."""
    L = omlib.load()
    rng = np.random.Generator(np.random.PCG64(123))
    x = np.concatenate([np.array([0.0, -0.0, 88.0, 88.8, 89.0, -87.0, -88.0, -103.0, -104.5, 100.5, 1e-30, -1e-30]),
                        rng.uniform(-30, 30, 800_000), rng.normal(0, 4, 800_000), rng.uniform(-110, 110, 400_000)]).astype(np.float32)
    xt = torch.from_numpy(x)
    xd = xt.to(dev)

    def run(func, inp, C=80):
        out = torch.empty_like(inp)
        omlib.check(L.om_ref_math(_p(inp), inp.numel(), func, C, _p(out), omlib.current_stream_ptr(dev)), "om_ref_math")
        return out.cpu().numpy()

    # one torch thread: with several, a thread's linear element range can start inside a row, which moves the boundary between
    # the row's vectorised part and its scalar tail (tools/gen_golden.py:single_thread)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    two = torch.stack([xt, xt], 1)                                   # column 0 is a strided view -> scalar loop
    got = run(2, xd)
    want = two[:, 0].sigmoid().numpy()
    bad = got.view(np.int32) != want.view(np.int32)
    assert not bad.any(), (int(bad.sum()), x[bad][:4], got[bad][:4], want[bad][:4])
    # rows of 64 contiguous values inside rows of 96: two full 32-lane steps per row, no scalar tail
    v = xt[:(x.size // 96) * 96].view(-1, 96)[:, :64]
    want = v.sigmoid().numpy()
    got = run(3, v.contiguous().to(dev)).reshape(want.shape)
    bad = got.view(np.int32) != want.view(np.int32)
    assert not bad.any(), (int(bad.sum()), v.numpy()[bad][:4], got[bad][:4], want[bad][:4])
    for C in (80, 20, 33, 96):                                       # postprocess.py:129: predict[..., 5:].sigmoid()
        rows = (x.size // (C + 5)) * (C + 5)
        t = xt[:rows].view(-1, C + 5)
        want = t[:, 5:].sigmoid().numpy()
        got = run(4, t[:, 5:].contiguous().to(dev), C).reshape(want.shape)
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), C
    # box sizes: MKL's vsExp is closed source; the device's correctly rounded expf is within one ulp of it
    sizes = torch.from_numpy(rng.uniform(-6, 6, 200_000).astype(np.float32))
    got = run(5, sizes.to(dev))
    assert _ulps(got, torch.stack([sizes, sizes], 1)[:, 0].exp().numpy()).max() <= 1
    torch.set_num_threads(nthreads)


def test_mask_predicate_sign_form(dev):
    """post_mask_kernel decides |d| < t for FINITE thresholds by the SIGN of the IEEE difference |d| - t (six vector instructions
    per pixel instead of nine).  For every pair below, incl. infinite and NaN d of both signs, zeros and denormals: sign set <=>
    numpy's abs(d) < t.  The one exception, which is why an infinite threshold takes the kernel's integer form: inf - inf is a NaN
    with the sign SET on this hardware."""
    L = omlib.load()
    inf, nan = np.inf, np.nan
    nnan = np.array([0xffc00001], dtype=np.uint32).view(np.float32)[0]
    den = np.float32(1e-45)
    d = np.array([0.5, -0.5, 0.25, -0.75, 0.0, -0.0, 0.0, inf, -inf, inf, -inf, nan, nnan, nan, nnan, 1.0, den, -den, 2 * den, 3.0e38,
                  -3.4028235e38, 1.0000001, 1.0, 0.99999994], dtype=np.float32)
    t = np.array([0.5, 0.5, 0.5, 0.5, 0.0, 0.0, den, inf, inf, 1.0, 1.0, 1.0, 1.0, inf, inf, inf, 2 * den, den, den, inf,
                  inf, 1.0, 1.0, 1.0], dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(9))
    d = np.concatenate([d, rng.normal(0, 1, 4096).astype(np.float32)])
    t = np.concatenate([t, np.abs(rng.normal(0, 1, 4096)).astype(np.float32)])
    t[-64:] = np.abs(d[-64:])                      # equal magnitudes: not inside
    xy = np.stack([d, t], 1).reshape(-1).copy()
    xd = torch.from_numpy(xy).to(dev)
    out = torch.empty_like(xd)
    omlib.check(L.om_ref_math(_p(xd), xd.numel(), 6, 1, _p(out), omlib.current_stream_ptr(dev)), "om_ref_math")
    bits = out.cpu().numpy().view(np.uint32)[0::2]
    with np.errstate(invalid="ignore"):
        want = np.abs(d) < t
    finite_t = np.isfinite(t) | ~np.isinf(d)            # every case but (|d| = inf, t = inf)
    got = bits >> 31 == 1
    assert np.array_equal(got[finite_t], want[finite_t]), (d[finite_t][got[finite_t] != want[finite_t]], t[finite_t][got[finite_t] != want[finite_t]])
    assert got[~finite_t].all() and not want[~finite_t].any() and (~finite_t).sum() == 2      # the documented exception


def test_postprocess_radix_path_skips_tiles_without_keys(dev):
    """conf_thresh = 1e-5 on clustered heads: ~34 000 pairs of an image pass (more than the compacted list holds, so the select kernel
    takes its three-pass radix path over the key array) while some 2048-pair tiles have no passing pair -- the decode kernel does
    not write those tiles' keys.  The workspace is first filled with the keys of an all-pass run (dense heads), so a tile read by
    mistake would inject stale keys.  Same candidates, classes, order and masks as the oracle."""
    size = (160, 192)
    pc = post_cfg(size)
    post = _hip_post(size, dev, conf_thresh=1e-5)
    dense = synth.synth_heads(105, 3, pc["grid_size"], regime="dense")
    post(tuple((b.to(dev), o.to(dev)) for b, o in dense))
    heads = synth.synth_heads(102, 3, pc["grid_size"], regime="clustered")
    oracle = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80, conf_thresh=1e-5)
    want = oracle(heads)
    got = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
    for b, (r, w) in enumerate(zip(got, want)):
        assert r["bbox"].shape[0] == w["bbox"].shape[0] > 0, b
        assert torch.equal(r["cls"].cpu(), w["cls"])
        assert torch.equal(post.last_keep[b].cpu().long(), w["keep"])
        _check_detections(r, w["bbox"].numpy(), w["cls"].numpy(), w["mask"].numpy(), ("radix path", b), exact_decode=True)


def test_postprocess_overflowed_box_width(dev):
    """A box whose width overflows to +inf (tw = 100 on the best candidate): its mask thresholds are infinite, which
    post_mask_kernel evaluates in the integer form (test_mask_predicate_sign_form); NMS sees infinite corners.  Same detections,
    same order, identical masks as the oracle; the infinite width reported as such."""
    size = (160, 192)
    pc = post_cfg(size)
    heads = synth.synth_heads(103, 2, pc["grid_size"], regime="sparse")
    bb = heads[2][0]
    t = bb[0].view(3, 85, *bb.shape[-2:])
    conf = torch.sigmoid(t[:, 5:]).amax(1) * torch.sigmoid(t[:, 4])
    a, y, x = np.unravel_index(int(conf.argmax()), conf.shape)
    t[a, 2, y, x] = 100.0
    oracle = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                                 conf_thresh=pc["conf_thresh"])
    want = oracle(heads)
    post = _hip_post(size, dev)
    got = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
    assert torch.isinf(want[0]["bbox"][:, 2]).sum() == 1
    for b, (r, w) in enumerate(zip(got, want)):
        gb, wb = r["bbox"].cpu().numpy(), w["bbox"].numpy()
        assert gb.shape == wb.shape and torch.equal(r["cls"].cpu(), w["cls"]) and torch.equal(post.last_keep[b].cpu().long(), w["keep"])
        assert np.array_equal(np.isinf(gb), np.isinf(wb))
        assert np.array_equal(gb[:, [0, 1, 4]], wb[:, [0, 1, 4]])
        fin = np.isfinite(wb[:, 2:4])
        assert _ulps(gb[:, 2:4][fin], wb[:, 2:4][fin]).max() <= 2
        assert torch.equal(r["mask"].cpu(), w["mask"].bool()), b
        k = int(np.flatnonzero(np.isinf(wb[:, 2]))[0]) if b == 0 else None
        if k is not None:
            assert w["mask"][k].any() and w["mask"][k].any(0).sum() > 1      # a mask that exists, wider than one column


def test_postprocess_full_size_batch_properties(dev):
    """bs=32 at 544x544 (BASELINE.json configs[2]): per-image results equal the single-image run
    bit for bit; counts bounded; masks only inside [0,1]."""
    size = (544, 544)
    pc = post_cfg(size)
    heads = synth.synth_heads(55, 32, pc["grid_size"], regime="mixed")
    dheads = _to_model_layout(heads, dev)
    post = _hip_post(size, dev)
    res = post(dheads)
    assert len(res) == 32
    for b in (0, 13, 31):
        one = post(tuple((h[0][b:b + 1], h[1][b:b + 1]) for h in dheads))[0]
        assert torch.equal(one["bbox"], res[b]["bbox"]) and torch.equal(one["cls"], res[b]["cls"])
        assert torch.equal(one["mask"], res[b]["mask"])
    for r in res:
        assert 0 < r["bbox"].shape[0] <= 100
        s = r["bbox"][:, 4]
        assert ((s > 0.005) & (s <= 1.0)).all()
        assert r["mask"].view(torch.uint8).max().item() <= 1


# ------------------------------------------------------------------------------------------------
# NMS
# ------------------------------------------------------------------------------------------------
def test_nms_known_answers(dev):
    from orienmask_amd.eval import batched_nms, nms
    kat = np.load(os.path.join(GOLDEN, "nms_kat.npz"))
    names = sorted(k[:-5] for k in kat.files if k.endswith("_keep"))
    for name in names:
        dets = torch.from_numpy(kat[name + "_dets"]).to(dev); cats = torch.from_numpy(kat[name + "_cats"]).to(dev)
        thr = float(kat[name + "_thr"])
        kd, kc, keep = batched_nms(dets, cats, threshold=thr)
        assert keep.cpu().numpy().tolist() == kat[name + "_keep"].tolist(), name
        assert kd.shape[0] == keep.shape[0] == kc.shape[0]
        _, _, keep2 = nms(dets, cats, threshold=thr)
        assert keep2.cpu().numpy().tolist() == kat[name + "_keep_plain"].tolist(), name


def test_nms_random_vs_oracle(dev):
    from orienmask_amd.eval import batched_nms
    rng = np.random.Generator(np.random.PCG64(9))
    for n in (2, 17, 128, 513, 1024):
        d = np.concatenate([rng.random((n, 2)), rng.random((n, 2)) * 0.3 + 0.02, rng.random((n, 1))], 1).astype(np.float32)
        c = rng.integers(0, 3, n)
        dt, ct = torch.from_numpy(d), torch.from_numpy(c)
        _, _, want = R.batched_nms(dt, ct, 0.45)
        _, _, got = batched_nms(dt.to(dev), ct.to(dev), threshold=0.45)
        assert got.cpu().tolist() == want.tolist(), n


def test_nms_cuda_backend_semantics(dev):
    """The reference's CUDA backend (eval/src/nms_kernel.cu, unbuildable today) restated in oracle/nms_cuda_ref.c: strict >,
    w*h areas, keep in score-descending order.  Known answers where the two backends differ, then random boxes."""
    from orienmask_amd.eval import batched_nms, nms
    # IoU exactly 0.5 (overlap 2 of union 4): the CPU backend suppresses (>=), the CUDA backend keeps both (>)
    d = torch.tensor([[1.5, 0.5, 3.0, 1.0, 0.9], [2.5, 0.5, 3.0, 1.0, 0.8], [8.0, 0.5, 3.0, 1.0, 0.7]])
    c = torch.zeros(3, dtype=torch.long)
    assert nms(d.to(dev), c.to(dev), 0.5, backend="cpu")[2].cpu().tolist() == [0, 2]
    assert nms(d.to(dev), c.to(dev), 0.5, backend="cuda")[2].cpu().tolist() == [0, 1, 2]
    # keep order: ascending index vs descending score
    d = torch.tensor([[0.2, 0.2, 0.1, 0.1, 0.3], [0.5, 0.5, 0.1, 0.1, 0.9], [0.8, 0.8, 0.1, 0.1, 0.6]])
    assert nms(d.to(dev), c.to(dev), 0.5, backend="cpu")[2].cpu().tolist() == [0, 1, 2]
    assert nms(d.to(dev), c.to(dev), 0.5, backend="cuda")[2].cpu().tolist() == [1, 2, 0]
    kat = np.load(os.path.join(GOLDEN, "nms_kat.npz"))
    for name in sorted(k[:-5] for k in kat.files if k.endswith("_keep")):
        dets = torch.from_numpy(kat[name + "_dets"]); cats = torch.from_numpy(kat[name + "_cats"])
        thr = float(kat[name + "_thr"])
        _, _, want = R.batched_nms(dets, cats, thr, backend="cuda")
        kd, kc, got = batched_nms(dets.to(dev), cats.to(dev), threshold=thr, backend="cuda")
        assert got.cpu().tolist() == want.tolist(), name
        assert torch.equal(kd.cpu(), dets[want]) and torch.equal(kc.cpu(), cats[want])
    rng = np.random.Generator(np.random.PCG64(10))
    for n in (2, 17, 128, 513, 1024, 1500):
        dnp = np.concatenate([rng.random((n, 2)), rng.random((n, 2)) * 0.3 + 0.02, rng.random((n, 1))], 1).astype(np.float32)
        dt, ct = torch.from_numpy(dnp), torch.from_numpy(rng.integers(0, 3, n))
        for normalized in (True, False):
            _, _, want = R.batched_nms(dt, ct, 0.45, normalized, backend="cuda")
            _, _, got = batched_nms(dt.to(dev), ct.to(dev), threshold=0.45, normalized=normalized, backend="cuda")
            assert got.cpu().tolist() == want.tolist(), (n, normalized)


def test_nms_beyond_one_workgroup(dev):
    """The reference's nms() has no size limit; om_nms_ex keeps everything that grows with n in the workspace (n <= 65536).
    4000 and 9000 boxes (63 and 141 mask words per row, several 64-row blocks per thread in the reduction), both backends."""
    from orienmask_amd.eval import nms
    rng = np.random.Generator(np.random.PCG64(11))
    for n in (4000, 9000):
        dnp = np.concatenate([rng.random((n, 2)), rng.random((n, 2)) * 0.08 + 0.01, rng.random((n, 1))], 1).astype(np.float32)
        dnp[0:n - 7:7, 4] = dnp[3:n - 4:7, 4]                          # score ties
        dt = torch.from_numpy(dnp)
        ct = torch.zeros(n, dtype=torch.long)
        assert nms(dt.to(dev), ct.to(dev), 0.3, backend="cpu")[2].cpu().tolist() == R.nms_cpu(dt, 0.3).tolist(), n
        assert nms(dt.to(dev), ct.to(dev), 0.3, backend="cuda")[2].cpu().tolist() == R.nms_cuda(dt, 0.3).tolist(), n


@pytest.mark.parametrize("backend,normalized,nms_pre", [("cuda", True, 400), ("cpu", False, 400), ("cuda", False, 400),
                                                        ("cpu", True, 1024), ("cuda", True, 700)])
@pytest.mark.parametrize("regime,seed", [("mixed", 201), ("sparse_many", 202), ("clustered", 203), ("ties_iou", 204)])
def test_postprocess_nms_options_match_oracle(dev, regime, seed, backend, normalized, nms_pre):
    """The fused postprocess with the reference's other NMS settings: its CUDA backend (what a GPU user of the reference
    gets, eval/function.py:98-101), batched_nms(normalized=False) (function.py:92) and nms_pre above 512 (the bit-matrix
    then lives in the workspace).  Same heads in, same indices / classes out as the oracle with those settings."""
    import functools
    from orienmask_amd.eval import batched_nms
    size = (544, 544) if regime == "ties_iou" else (160, 192)
    pc = post_cfg(size)
    heads = synth.synth_heads(seed, 2, pc["grid_size"], regime=regime)
    oracle = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                                 conf_thresh=pc["conf_thresh"], nms_pre=nms_pre, nms_backend=backend, nms_normalized=normalized)
    want = oracle(heads)
    post = _hip_post(size, dev, nms_pre=nms_pre,
                     nms_func=functools.partial(batched_nms, threshold=0.5, normalized=normalized, backend=backend))
    assert post.nms_backend == backend
    got = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
    for b, (r, w) in enumerate(zip(got, want)):
        assert torch.equal(post.last_keep[b].cpu().long(), w["keep"]), (regime, backend, b)
        _check_detections(r, w["bbox"].numpy(), w["cls"].numpy(), w["mask"].numpy(), (regime, backend, normalized, b),
                          exact_decode=True)


# ------------------------------------------------------------------------------------------------
# preprocess
# ------------------------------------------------------------------------------------------------
def test_preprocess_matches_reference_golden(dev):
    """om_preprocess (fused permute + bilinear resize + normalise + pad) vs the reference's own
    FastCOCOTransform + infer.pad outputs: bit-exact (float ops in torch's order)."""
    from orienmask_amd.transform import FastCOCOTransform, build_transform, pad
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    names = sorted(k[:-5] for k in g.files if k.endswith("_seed"))
    for name in names:
        n, h, w = (int(v) for v in g[name + "_shape"])
        size = tuple(int(v) for v in g[name + "_size"])
        img = synth.synth_photo_batch(int(g[name + "_seed"]), n, h, w).to(dev)
        pipeline = [dict(type="Normalize", mean=(0, 0, 0), std=(255, 255, 255))]
        if size != (h, w):
            pipeline.insert(0, dict(type="Resize", size=size, interpolation="bilinear", align_corners=False))
        tf = build_transform(dict(type="FastCOCOTransform", pipeline=pipeline, use_cuda=True))
        assert isinstance(tf, FastCOCOTransform)
        fused, info = tf.padded(img)                       # one kernel
        two_step, info2 = pad(tf(img))                     # transform, then pad (the reference's call sequence)
        assert info == info2 == g[name + "_pad"].tolist(), name
        assert list(fused.shape) == g[name + "_outshape"].tolist(), name
        assert torch.equal(fused, two_step), name
        flat = fused.cpu().reshape(-1)
        got = flat[torch.from_numpy(g[name + "_idx"])].numpy()
        assert np.array_equal(got, g[name + "_samples"]), (name, np.abs(got - g[name + "_samples"]).max())
        assert abs(flat.double().sum().item() - g[name + "_sum"][0]) <= 1e-9 * g[name + "_sum"][1]
        if name + "_out" in g.files:
            assert np.array_equal(fused.cpu().numpy(), g[name + "_out"]), name


def test_preprocess_feeds_the_model(dev):
    """infer.py's sequence on the HIP path: transform -> pad -> model -> postprocess, arbitrary image size."""
    from orienmask_amd.transform import FastCOCOTransform
    img = synth.synth_photo_batch(77, 1, 300, 420).to(dev)
    tf = FastCOCOTransform([FastCOCOTransform.ShortEdgeResize(160, 256), FastCOCOTransform.Normalize((0, 0, 0), (255, 255, 255))])
    x, info = tf.padded(img)
    assert x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0 and info[4:] == list(x.shape[2:])
    want = R.pad_to_divisor(R.fast_coco_transform(img.cpu(), (160, 224)), 32, 0)[0]
    assert torch.equal(x.cpu(), want)
    sd = synth.synth_state_dict(5, obj_bias=-16.0, head_gain=4.0)
    net = _hip_model(sd, dev)
    with torch.no_grad():
        out = net(x)
    ref = R.forward(sd, want)
    for (gb, go), (rb, ro) in zip(out, ref):
        assert _rel_err(gb.cpu(), rb) < REL_TOL and _rel_err(go.cpu(), ro) < REL_TOL


def test_library_loaded_before_torch_still_works(dev):
    """A process that loads liborienmask_hip.so before it imports torch (as __graft_entry__.build() followed by smoke() does)
    must end up with one HIP runtime: lib.load() imports torch first.  Runs in a fresh interpreter."""
    import subprocess, sys
    code = ("from orienmask_amd import lib; lib.load(); import __graft_entry__ as g; g.smoke(); print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_tester_and_infer_loops(dev):
    """The reference's two callers on the HIP path: Tester.test (tester.py:26-62) and the infer.py loop."""
    from orienmask_amd.tester import SyntheticLoader, Tester, infer_loop
    from orienmask_amd.transform import FastCOCOTransform
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net = _hip_model(sd, dev)
    post = _hip_post((544, 544), dev)
    seen = []
    tester = Tester(net, post, SyntheticLoader(6, 4, seed=500), dev, on_batch=lambda info, dets: seen.append((info, dets)))
    stats = tester.test(verbose=False)
    assert set(stats) >= {"Network Forward", "Postprocess", "Convert Format", "detections"}
    assert stats["Network Forward"]["fps"] > 0 and len(seen) == 2 and [len(s[1]) for s in seen] == [4, 2]
    assert [i["id"] for s in seen for i in s[0]] == list(range(6))
    # the same loader with two batches in flight: same detections in the same order, one combined timer
    seen2 = []
    tester2 = Tester(net, post, SyntheticLoader(6, 4, seed=500), dev,
                     on_batch=lambda info, dets: seen2.append((info, [{k: v.clone() for k, v in d.items()} for d in dets])))
    stats2 = tester2.test(verbose=False, in_flight=2)
    assert set(stats2) == {"Forward & Postprocess", "Convert Format", "detections"} and stats2["detections"] == stats["detections"]
    assert [i["id"] for s_ in seen2 for i in s_[0]] == list(range(6))
    for (_, a), (_, b) in zip(seen, seen2):
        for da, db in zip(a, b):
            assert torch.equal(da["bbox"], db["bbox"]) and torch.equal(da["cls"], db["cls"]) and torch.equal(da["mask"], db["mask"])
    tf = FastCOCOTransform([FastCOCOTransform.Resize((544, 544)), FastCOCOTransform.Normalize((0, 0, 0), (255, 255, 255))])
    imgs = [synth.synth_photo_batch(900 + i, 1, 240 + 16 * i, 320)[0] for i in range(3)]
    dets, pads, log = infer_loop(net, tf, post, imgs, dev, warmup=2)
    assert len(dets) == 3 and all(p == [0, 0, 0, 0, 544, 544] for p in pads)
    assert set(log) == {"Main Loop", "Load data", "Forward & Postprocess"}
    # same image, same answer as the direct call sequence
    x, _ = tf.padded(imgs[1].to(dev).unsqueeze(0))
    with torch.no_grad():
        again = post(net(x))[0]
    assert torch.equal(again["bbox"], dets[1]["bbox"]) and torch.equal(again["mask"], dets[1]["mask"])
    # the loop above replayed ONE captured hipGraph (the default for a fixed shape); eager launches give the same detections
    dets_e, _, _ = infer_loop(net, tf, post, imgs, dev, warmup=0, use_graph=False)
    for a, b in zip(dets, dets_e):
        assert torch.equal(a["bbox"], b["bbox"]) and torch.equal(a["cls"], b["cls"]) and torch.equal(a["mask"], b["mask"])
    # the reference's contract for `postprocess` is a callable (infer.py:154-156): a plain function, and a postprocess with a foreign
    # nms_func (host work in the middle of the step), cannot be captured -- the loop runs them eagerly instead of failing
    dets_f, _, _ = infer_loop(net, tf, lambda pred: post(pred), imgs, dev, warmup=1)
    from orienmask_amd.eval import batched_nms
    import functools
    post_foreign = _hip_post((544, 544), dev, nms_func=functools.partial(batched_nms, threshold=0.5))
    dets_g, _, _ = infer_loop(net, tf, post_foreign, imgs, dev, warmup=1)
    for a, b, c in zip(dets, dets_f, dets_g):
        assert torch.equal(a["bbox"], b["bbox"]) and torch.equal(a["mask"], b["mask"])
        assert torch.equal(a["bbox"], c["bbox"]) and torch.equal(a["cls"], c["cls"]) and torch.equal(a["mask"], c["mask"])
    # latency mode is restored when the loop throws
    cells = net.latency_cells

    def boom(pred):
        raise ValueError("postprocess failed")
    with pytest.raises(ValueError):
        infer_loop(net, tf, boom, imgs, dev, warmup=0)
    assert net.latency_cells == cells


def test_pack_on_device_equals_pack_on_host(dev):
    """orienmask_amd.pack computes where the weights are (a model on the GPU packs on the GPU: model load 12 s -> under 1 s);
    its arithmetic is fixed-order float64 elementwise work, so the three blobs are the same BITS from either device -- what a
    checkpoint packed on a CPU-only host and one packed on the GPU box must be for RCCL-broadcast blobs to be interchangeable."""
    from orienmask_amd import lib as _lib, pack
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    sd = synth.synth_state_dict(11, obj_bias=-3.0, head_gain=2.0)
    net = OrienMaskYOLOFPNPlus(3, 80).eval()
    h = net._ensure_handle(); L = _lib.load()
    sd_dev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sd.items()}
    for name, total in (("pack_state_dict", L.om_model_weight_floats(h)), ("pack_state_dict_split", L.om_model_weight_split_words(h)),
                        ("pack_state_dict_f16", L.om_model_weight_halfs(h))):
        fn = getattr(pack, name)
        fn = getattr(fn, "__wrapped__", fn)                    # not the test session's memo (conftest.py)
        host, device = fn(sd, net._layers, total), fn(sd_dev, net._layers, total)
        assert host.device.type == "cpu" and device.device.type == "cuda"
        bits = torch.int32 if host.dtype == torch.float32 else torch.int16
        assert torch.equal(host.view(bits), device.cpu().view(bits)), name


def test_build_tester_from_checkpoint_file(dev, tmp_path):
    """test.py's build_tester(config, checkpoint) on a reference-format .pth file."""
    from orienmask_amd import builder
    from orienmask_amd.tester import SyntheticLoader
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    cfg = dict(model=dict(type="OrienMaskYOLOFPNPlus", num_anchors=3, num_classes=80, pretrained="checkpoints/pretrained.pth",
                          freeze_backbone=False, backbone_batchnorm_eval=False))
    path = str(tmp_path / "epoch100.pth")
    torch.save(dict(epoch=100, state_dict=sd, optimizer={}, lr_scheduler={}, monitor_best=0.3, config=cfg), path)
    test_cfg = dict(postprocess=dict(type="OrienMaskYOLOPostProcess", nms=dict(type="batched_nms", threshold=0.5),
                                     **post_cfg((544, 544))))
    tester = builder.build_tester(test_cfg, path, SyntheticLoader(2, 2, seed=700), device=dev)
    stats = tester.test(verbose=False)
    assert stats["detections"] > 0
    x = synth.synth_image_batch(700, 2, 544, 544).to(dev)
    assert tester.model.precision == "f32_split"               # a checkpoint's config without `precision` gets the plugin default
    with torch.no_grad():
        direct = _hip_model(sd, dev, tester.model.precision)(x)
        via_ckpt = tester.model(x)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(direct, via_ckpt))


# ------------------------------------------------------------------------------------------------
# COCO-format conversion
# ------------------------------------------------------------------------------------------------
def test_coco_format_matches_reference_golden(dev):
    """om_recover_bbox / om_recover_masks_rle vs the reference's _recover_shape_bbox / _recover_shape_segm
    outputs (bit-exact boxes and resized masks) and the oracle's run lengths / RLE strings."""
    from orienmask_amd import coco_format as CF
    from test_oracle_golden import _coco_cases
    for name, info, masks, bbox, xywh, seg in _coco_cases():
        got_xywh = CF.recover_shape_bbox(torch.from_numpy(bbox).to(dev), info).cpu().numpy()
        assert np.array_equal(got_xywh, xywh), (name, np.abs(got_xywh - xywh).max())
        rles, resized = CF.recover_masks_rle(torch.from_numpy(masks).to(dev), info, return_resized=True)
        assert np.array_equal(resized.cpu().numpy().astype(bool), seg), name
        for k, rle in enumerate(rles):
            want_counts = R.rle_counts(seg[k])
            assert rle["size"] == [info["height"], info["width"]]
            assert rle["counts"] == R.rle_to_string(want_counts), (name, k)
            assert R.rle_string_decode(rle["counts"], seg[k].size) == want_counts
    # a mask with more runs than the first buffer guess takes the retry path
    noisy = (torch.rand(1, 64, 64, generator=torch.Generator().manual_seed(1)) < 0.5)
    info = dict(height=64, width=64)
    rles = CF.recover_masks_rle(noisy.to(dev), info, max_runs=64)
    assert rles[0]["counts"] == R.rle_to_string(R.rle_counts(R.recover_shape_segm(noisy, info)[0].numpy()))


def test_coco_formatter_end_to_end(dev):
    """Detections of the HIP postprocess -> COCO result dicts, vs the oracle on the same detections."""
    from orienmask_amd.coco_format import COCOFormatter
    pc = post_cfg((96, 128))
    heads = synth.synth_heads(321, 2, pc["grid_size"], regime="mixed")
    post = _hip_post((96, 128), dev)
    dets = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
    infos = [dict(id=11, height=120, width=160, collate_pad=[0, 0, 0, 0, 96, 128]),
             dict(id=12, height=75, width=100, pad=[2, 2, 4, 4, 96, 128], hflip=True)]
    cat2label = list(range(1, 81))
    res = COCOFormatter(cat2label, with_mask=True).to_coco_format(infos, dets)
    assert len(res["bbox"]) == len(res["segm"]) == sum(int(d["bbox"].shape[0]) for d in dets)
    i = 0
    for info, d in zip(infos, dets):
        xywh = R.recover_shape_bbox(d["bbox"].cpu(), info)
        seg = R.recover_shape_segm(d["mask"].cpu(), info).numpy()
        for k in range(d["bbox"].shape[0]):
            b, s = res["bbox"][i], res["segm"][i]
            assert b["image_id"] == s["image_id"] == info["id"] and b["category_id"] == cat2label[int(d["cls"][k])]
            assert b["bbox"] == xywh[k].tolist() and abs(b["score"] - float(d["bbox"][k, 4])) == 0
            assert s["segmentation"] == {"size": [info["height"], info["width"]],
                                         "counts": R.rle_to_string(R.rle_counts(seg[k]))}
            i += 1


@pytest.mark.parametrize("batch,prec", [(1, "f32"), (4, "f32"), (2, "f16"), (6, "f32_split")])
def test_graphed_pipeline_matches_eager(dev, batch, prec):
    """hipGraph replay of forward + postprocess == the eager call sequence, bit for bit.  Replays run back to back
    on fresh inputs with no eager call on the same workspaces in between (an eager call re-clears the tile-queue
    tickets and the radix histograms, which once hid a replay that did not), and the eager side is a second
    model / postprocess instance."""
    from orienmask_amd.graph import GraphedPipeline
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net, net_ref = _hip_model(sd, dev).set_precision(prec), _hip_model(sd, dev).set_precision(prec)
    post, post_ref = _hip_post((544, 544), dev), _hip_post((544, 544), dev)
    xs = [synth.synth_image_batch(801 + i, batch, 544, 544).to(dev) for i in range(3)]
    pipe = GraphedPipeline(net, post, xs[0])
    got = []
    for x in (xs[0], xs[1], xs[2], xs[1]):
        got.append([{k: v.clone() for k, v in d.items()} for d in pipe(x)])
    for x, g_list in zip((xs[0], xs[1], xs[2], xs[1]), got):
        with torch.no_grad():
            want = post_ref(net_ref(x))
        assert len(g_list) == len(want) == batch
        for g, w in zip(g_list, want):
            assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"])
    with pytest.raises(ValueError):
        pipe(torch.zeros(batch + 1, 3, 544, 544, device=dev))
    if batch == 1 and prec == "f32":
        # an eager call at another shape drops the model's / postprocess' cached workspaces; the pipeline holds its own
        # references to what the captured kernels point into, so the replay stays valid
        with torch.no_grad():
            _hip_post((96, 96), dev)(net(torch.rand(2, 3, 96, 96, device=dev)))      # drops the model's 544x544 workspace
            post(net_ref(xs[0][:1].repeat(3, 1, 1, 1)))                              # ... and the postprocess' (other batch)
            filler = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(64)]    # would land in a freed block
            again = [{k: v.clone() for k, v in d.items()} for d in pipe(xs[2])]
            want = post_ref(net_ref(xs[2]))
        del filler
        assert all(torch.equal(g[k], w[k]) for g, w in zip(again, want) for k in ("bbox", "cls", "mask"))
        # new weights after capture: replay must refuse, not run the stale blobs
        net.load_state_dict(synth.synth_state_dict(4, obj_bias=-16.0, head_gain=4.0), strict=True)
        with torch.no_grad():
            net(xs[0])
        with pytest.raises(RuntimeError):
            pipe(xs[0])


def test_graphed_pipeline_with_the_wide_3x3_form(dev):
    """The two-kernel wide form of the 1/32-scale 3x3 layers (round 6: a pre-pass + a consumer kernel per layer, its transformed
    input in a per-layer scratch of the workspace) inside a captured hipGraph: 140 strips of 32 x 544 (enough tiles for om_forward to
    choose the form) replayed twice on fresh inputs == the eager call sequence of a second model instance, bit for bit."""
    from orienmask_amd.graph import GraphedPipeline
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net, net_ref = _hip_model(sd, dev, "f32_split"), _hip_model(sd, dev, "f32_split")
    assert any(v.startswith("wino14_wide_kernel") for _, v in net.layer_kernels(140, 32, 544))
    post, post_ref = _hip_post((32, 544), dev), _hip_post((32, 544), dev)
    xs = [synth.synth_image_batch(811 + i, 140, 32, 544).to(dev) for i in range(2)]
    pipe = GraphedPipeline(net, post, xs[0])
    got = [[{k: v.clone() for k, v in d.items()} for d in pipe(x)] for x in (xs[0], xs[1], xs[0])]
    for x, g_list in zip((xs[0], xs[1], xs[0]), got):
        with torch.no_grad():
            want = post_ref(net_ref(x))
        assert len(g_list) == len(want) == 140
        for g, w in zip(g_list, want):
            assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"])


# ------------------------------------------------------------------------------------------------
# fp16-activation configuration (BASELINE.json configs[4]).  The reference has no reduced-precision path: the
# arithmetic is defined by oracle.forward_f16 (fp16 operands, fp32 sums, one rounding per layer) -- "parity
# unpinned" for this configuration; what is checked is HIP == that definition, and closeness to the fp32 path.
# ------------------------------------------------------------------------------------------------
F16_CONV_CASES = [
    # B, H, W, cin, cout, k, stride, leaky, residual, out_f32
    (2, 16, 16, 32, 64, 3, 2, 1, False, 0),      # cin = 32: two taps per k-step (conv2.0)
    (1, 18, 14, 32, 64, 3, 1, 1, True, 0),       # conv2.1.conv.1
    (2, 12, 20, 64, 32, 1, 1, 1, False, 0),      # one k-step
    (3, 8, 8, 128, 128, 3, 1, 1, True, 0),
    (1, 34, 34, 128, 256, 3, 1, 1, False, 0),
    (2, 20, 20, 256, 128, 1, 1, 1, False, 0),
    (2, 16, 16, 64, 128, 3, 2, 1, False, 0),
    (2, 9, 7, 256, 255, 1, 1, 0, False, 1),      # box head: fp32 out, cout not a multiple of 8
    (5, 4, 4, 1024, 512, 1, 1, 1, False, 0),
    (4, 32, 32, 128, 256, 3, 1, 1, True, 0),     # enough rows for the 256 x 128 tile
    (2, 5, 3, 64, 128, 3, 1, 1, False, 0),       # rows shorter than the kernel: every pixel touches padding
    (3, 1, 7, 64, 128, 3, 1, 1, True, 0),        # one-row images: the raster neighbours above/below are other images
    (2, 7, 1, 64, 64, 3, 1, 0, False, 0),        # one-column images
    (9, 17, 17, 512, 1024, 3, 1, 1, True, 0),    # conv6-like: several images per tile
    (2, 24, 40, 256, 128, 3, 1, 1, False, 1),    # fp32 output from the shared-patch kernel
    (2, 136, 136, 32, 128, 3, 1, 1, True, 0),    # tall-patch kernel: seven patch pieces (W = 136), one chunk
    (3, 68, 68, 64, 128, 3, 1, 1, False, 0),     # ... six pieces, two chunks (the patch double buffer)
    (1, 40, 190, 96, 256, 3, 1, 0, True, 0),     # ... the longest rows it takes, three chunks, two N tiles
    (1, 8, 200, 32, 128, 3, 1, 1, False, 0),     # rows too long for it: the shared-patch kernel whatever the variant
]


@pytest.mark.parametrize("case,variant", [(c, v) for c in F16_CONV_CASES for v in ((0, 2) if c[5] == 3 and c[6] == 1 else (0,))])
def test_conv_f16_layer_matches_torch(dev, case, variant):
    """variant: om_set_conv3x3_f16_variant -- 0 the shared-patch 3x3 kernel of rounds 1-4, 2 the tall-patch kernel wherever it
    can run (the default, 1, is one of the two per layer)."""
    from orienmask_amd.pack import conv_weights_f16
    B, H, W, cin, cout, k, stride, leaky, use_res, out_f32 = case
    L = omlib.load()
    previous = L.om_get_conv3x3_f16_variant()       # OM_C3_TALL of the environment, or the default: restored below
    omlib.check(L.om_set_conv3x3_f16_variant(variant), "om_set_conv3x3_f16_variant")
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(B, cin, H, W, generator=g).half()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half()
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.2
    Ho, Wo = H // stride, W // stride
    res = torch.randn(B, cout, Ho, Wo, generator=g).half() if use_res else None
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, stride, k // 2)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if leaky:
        want = torch.where(want > 0, want, want * 0.1)
    if use_res:
        want = want + res.double()
    cpad = (cout + 31) // 32 * 32
    wd = conv_weights_f16(w.float(), cpad).contiguous().to(dev)
    sp = torch.zeros(cpad); sp[:cout] = scale
    hp = torch.zeros(cpad); hp[:cout] = shift
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    sd_, hd = sp.to(dev), hp.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    ostride = 256 if out_f32 else cout
    out = torch.full((B, Ho, Wo, ostride), float("nan"), device=dev, dtype=torch.float32 if out_f32 else torch.float16)
    try:
        rc = L.om_conv2d_f16(_p(xd), B, H, W, cin, cin, _p(wd), _p(sd_), _p(hd), cout, k, stride, leaky,
                             _p(rd) if use_res else None, cout if use_res else 0, _p(out), ostride, out_f32,
                             omlib.current_stream_ptr(dev))
    finally:
        L.om_set_conv3x3_f16_variant(previous)
    omlib.check(rc, "om_conv2d_f16")
    got = out[..., :cout].cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    if out_f32:
        assert _rel_err(got, want) < 5e-6, case
    else:
        # one rounding to fp16 (half an ulp = 2^-11 relative) of an fp32-accumulated value (a few 1e-6 of the scale)
        bound = want.abs() * 2.0 ** -10 + 4e-6 * want.abs().max()
        assert ((got - want).abs() <= bound).all(), (case, ((got - want).abs() / bound).max().item())
        assert (got == want.half().double()).double().mean() > 0.98


def test_stem_f16_matches_torch(dev):
    L = omlib.load()
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 40, 72
    x = torch.rand(B, 3, H, W, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.2
    want = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.1)
    wp = w.permute(0, 2, 3, 1).reshape(32, 27).contiguous().to(dev)
    out = torch.full((B, H, W, 32), float("nan"), device=dev, dtype=torch.float16)
    xd, scd, shd = x.to(dev), scale.to(dev), shift.to(dev)
    rc = L.om_conv2d_stem_f16(_p(xd), B, H, W, _p(wp), _p(scd), _p(shd), 32, _p(out), omlib.current_stream_ptr(dev))
    omlib.check(rc, "om_conv2d_stem_f16")
    got = out.cpu().permute(0, 3, 1, 2).double()
    bound = want.abs() * 2.0 ** -10 + 4e-6 * want.abs().max()
    assert ((got - want).abs() <= bound).all()


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 34, 50), (3, 16, 32), (1, 96, 160)])
def test_stem2_f16_matches_two_kernels(dev, shape):
    """conv_stem2_f16_kernel -- backbone.conv1 + backbone.conv2.0 of the fp16-activation configuration as ONE kernel (round 5) --
    against the definition of that configuration (float64 conv1, ONE rounding to fp16, fp16 weights, float64 conv2.0, one rounding)
    and against om_conv2d_stem_f16 followed by om_conv2d_f16; partial tiles in both directions, an output view with its own stride."""
    from orienmask_amd.pack import conv_weights_f16
    B, H, W = shape
    L = omlib.load()
    g = torch.Generator().manual_seed(sum(shape) + 7)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 0.5
    w1 = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    sc1 = torch.rand(32, generator=g) + 0.5
    sh1 = torch.randn(32, generator=g) * 0.2
    w2 = (torch.randn(64, 32, 3, 3, generator=g) / (32 * 9) ** 0.5).half()
    sc2 = torch.rand(64, generator=g) + 0.5
    sh2 = torch.randn(64, generator=g) * 0.2
    a = torch.nn.functional.conv2d(x.double(), w1.double(), None, 1, 1) * sc1.double().view(1, -1, 1, 1) + sh1.double().view(1, -1, 1, 1)
    a = torch.where(a > 0, a, a * 0.1).half().double()
    want = torch.nn.functional.conv2d(a, w2.double(), None, 2, 1) * sc2.double().view(1, -1, 1, 1) + sh2.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.1)
    w1p = w1.permute(0, 2, 3, 1).reshape(32, 27).contiguous().to(dev)
    w2d = conv_weights_f16(w2.float(), 64).contiguous().to(dev)
    xd, sc1d, sh1d, sc2d, sh2d = x.to(dev), sc1.to(dev), sh1.to(dev), sc2.to(dev), sh2.to(dev)
    st = omlib.current_stream_ptr(dev)
    Ho, Wo = H // 2, W // 2
    mid = torch.empty(B, H, W, 32, device=dev, dtype=torch.float16)
    omlib.check(L.om_conv2d_stem_f16(_p(xd), B, H, W, _p(w1p), _p(sc1d), _p(sh1d), 32, _p(mid), st), "om_conv2d_stem_f16")
    two = torch.full((B, Ho, Wo, 64), float("nan"), device=dev, dtype=torch.float16)
    omlib.check(L.om_conv2d_f16(_p(mid), B, H, W, 32, 32, _p(w2d), _p(sc2d), _p(sh2d), 64, 3, 2, 1, None, 0, _p(two), 64, 0, st), "om_conv2d_f16")
    ostride = 80
    buf = torch.full((B, Ho, Wo, ostride), float("nan"), device=dev, dtype=torch.float16)
    view = buf[..., 8:]
    omlib.check(L.om_conv2d_stem2_f16(_p(xd), B, H, W, _p(w1p), _p(sc1d), _p(sh1d), _p(w2d), _p(sc2d), _p(sh2d), 64, 1,
                                      ctypes.c_void_p(view.data_ptr()), ostride, st), "om_conv2d_stem2_f16")
    one = buf[..., 8:72]
    assert torch.isnan(buf[..., :8]).all() and torch.isnan(buf[..., 72:]).all()      # nothing outside the 64 channels
    got = one.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    # one rounding to fp16 of a value whose conv1 operands may differ from the float64 ones by a rounding tie (rare): the same
    # bound as the single fp16 layers, plus a few 1e-4 of scale for those
    bound = want.abs() * 2.0 ** -10 + 3e-4 * want.abs().max()
    assert ((got - want).abs() <= bound).all(), ((got - want).abs() / bound).max().item()
    same = (one == two).float().mean().item()
    print("stem2 f16 %s: %.4f of the outputs equal the two-kernel path's bit for bit" % (shape, same))
    assert same > 0.97 and _rel_err(one.cpu().double(), two.cpu().double()) < 2e-3


F16_FWD_TOL = 1e-2       # of each head tensor's scale, HIP fp16 path vs oracle.forward_f16 (same arithmetic, other sum order)
F16_VS_F32_TOL = 5e-2    # fp16 path vs the fp32 path on the same weights and image


@pytest.mark.parametrize("batch,size", [(2, (544, 544)), (1, (320, 416))])
def test_forward_f16_matches_oracle(dev, batch, size):
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(21, batch, size[0], size[1])
    net = _hip_model(sd, dev).set_precision("f16")
    with torch.no_grad():
        out = net(x.to(dev))
    torch.cuda.synchronize()
    want = R.forward_f16(sd, x)
    want32 = R.forward(sd, x)
    worst = 0.0
    for (gb, go), (wb, wo), (fb, fo) in zip(out, want, want32):
        assert gb.dtype == torch.float32 and go.dtype == torch.float32
        for g, w, f in ((gb, wb, fb), (go, wo, fo)):
            g = g.cpu()
            assert torch.isfinite(g).all()
            worst = max(worst, _rel_err(g, w))
            assert _rel_err(g, w) < F16_FWD_TOL
            assert _rel_err(g, f) < F16_VS_F32_TOL
    print("fp16 forward vs oracle.forward_f16: worst rel err %.3e" % worst)


@pytest.mark.parametrize("precision", ["f32", "f32_split"])
def test_forward_bs6_uses_f24_and_matches_oracle(dev, precision):
    """With fp32 operands, from 1700 1/32-scale cells on (bs >= 6 at 544x544) the stride-1 3x3 layers run Winograd F(2x4,3x3);
    below, F(2x2,3x3); in the default split-operand precision they run the fused F(4,3) kernel at every batch size.  Head tensors
    of a 6-image batch against the oracle, and the kernel choice itself as om_layer_tile reports it."""
    sd = synth.synth_state_dict(9, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(26, 6, 544, 544)
    net = _hip_model(sd, dev, precision)
    with torch.no_grad():
        out = net(x.to(dev))
    torch.cuda.synchronize()
    assert out.flags() == 0
    k6 = dict(net.layer_kernels(6, 544, 544)); k2 = dict(net.layer_kernels(2, 544, 544))
    if precision == "f32":
        assert k6["orien_head.2"].startswith("wino24_gemm") and k2["orien_head.2"].startswith("wino_gemm")
        assert k6["backbone.conv6.2.conv.1"].startswith("wino24_gemm") and k2["backbone.conv6.2.conv.1"].startswith("wino_gemm")
    else:
        for k in (k6, k2):
            assert k["orien_head.2"].startswith("wino14_split") and k["backbone.conv6.2.conv.1"].startswith("wino14_split")
    ref = _oracle_once(("fwd", 9, 26, 6, 544, 544), lambda: R.forward(sd, x))
    for (gb, go), (rb, ro) in zip(out, ref):
        assert _rel_err(gb.cpu(), rb) < REL_TOL and _rel_err(go.cpu(), ro) < REL_TOL


@pytest.mark.parametrize("precision", ["f32", "f32_split"])
def test_headline_bs32_forward_and_postprocess(dev, precision):
    """BASELINE configs[2] at its own size: forward + postprocess of 32 x 544x544 in one call (the bench's architecture and
    input; head gains chosen so that scores do not saturate into exact ties).
    Three images of the batch (first, middle, last) against the CPU oracle: head tensors within 1e-4 of scale, detections
    index-exact against the oracle's own end-to-end run AND bit-exact decode against the oracle's postprocess of the HIP
    heads; and the whole batch bit-identical to a batch of 8 holding the same three images (both sizes run the Winograd
    F(2x4,3x3) path; tile shapes, grid sizes and the workspace layout differ)."""
    sd = synth.synth_state_dict(3, obj_bias=-3.0, head_gain=0.7)      # unsaturated heads: no exact score ties (gen_golden.py)
    x = synth.synth_image_batch(1000, 32, 544, 544)
    net = _hip_model(sd, dev, precision)
    if precision == "f32_split":
        assert dict(net.layer_kernels(32, 544, 544))["orien_head.2"].startswith("wino14_split_kernel")
    post = _hip_post((544, 544), dev)
    pick = [0, 15, 31]
    with torch.no_grad():
        out = net(x.to(dev))
        res = post(out)
        heads32 = [(b[pick].clone(), o[pick].clone()) for b, o in out]
        dets32 = [{k: v.clone() for k, v in res[i].items()} for i in pick]
        assert len(res) == 32 and all(0 < r["bbox"].shape[0] <= 100 for r in res)
        x8 = x[pick + [1, 2, 3, 4, 5]]
        out8 = net(x8.to(dev))
        res8 = post(out8)
    for (b32, o32), (b8, o8) in zip(heads32, out8):
        assert torch.equal(b32, b8[:3]) and torch.equal(o32, o8[:3])
    for d32, d8 in zip(dets32, res8[:3]):
        assert all(torch.equal(d32[k], d8[k]) for k in ("bbox", "cls", "mask"))
    pc = post_cfg((544, 544))
    oracle_post = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                                      conf_thresh=pc["conf_thresh"])
    if "headline" not in _ORACLE_CACHE:                 # the CPU oracle's forward + postprocess once for both precisions
        ref_ = R.forward(sd, x[pick])
        _ORACLE_CACHE["headline"] = (ref_, oracle_post(ref_))
    ref, want_e2e = _ORACLE_CACHE["headline"]                                     # the reference's end-to-end answer
    for (gb, go), (rb, ro) in zip(heads32, ref):
        assert _rel_err(gb.cpu(), rb) < REL_TOL and _rel_err(go.cpu(), ro) < REL_TOL
    want_same_heads = oracle_post([(b.cpu(), o.cpu()) for b, o in heads32])       # same heads in: decode must be bit-exact
    for i, d in enumerate(dets32):
        _check_detections(d, want_same_heads[i]["bbox"].numpy(), want_same_heads[i]["cls"].numpy(),
                          want_same_heads[i]["mask"].numpy(), ("bs32 same heads", pick[i]), exact_decode=True)
        _check_detections_composed(d, want_e2e[i]["bbox"].numpy(), want_e2e[i]["cls"].numpy(), want_e2e[i]["mask"].numpy(),
                                   ("bs32 end to end", precision, pick[i]),
                                   margin_ctx=_margin_ctx(oracle_post, [(b.cpu(), o.cpu()) for b, o in heads32], i))


def _sorted_within_ties(bbox, cls, mask):
    """Detections reordered inside groups of bit-identical scores (by class, then box bytes): torch.topk / sort leave the order
    of exact ties unspecified, so a tie group is compared as a set; everything else keeps its position."""
    bbox = np.ascontiguousarray(bbox, dtype=np.float32)
    key = [(-int(np.float32(bbox[i, 4]).view(np.int32)), int(cls[i]), bbox[i, :4].tobytes()) for i in range(bbox.shape[0])]
    order = sorted(range(bbox.shape[0]), key=lambda i: key[i])
    return bbox[order], np.asarray(cls)[order], np.asarray(mask)[order]


def test_bench_workload_bs32_detections(dev):
    """bench.py's own workload (config.workload of the bench line): seed-3 weights with obj_bias -16 / head_gain 4, 32 x 544^2,
    the plugin's DEFAULT precision, one call.  These heads saturate the sigmoids -- many scores are exactly 1.0 -- so three
    images of the batch are checked against the CPU oracle with exact-tie groups compared as sets: (a) the oracle's postprocess
    on the SAME (HIP) heads: scores and box centres bit-identical, classes and masks identical, group by group; (b) head
    tensors within 1e-4 of the oracle's forward; (c) the composed path against the oracle's own end-to-end detections."""
    import bench
    sd = synth.synth_state_dict(bench.WEIGHT_SEED, obj_bias=bench.OBJ_BIAS, head_gain=bench.HEAD_GAIN)
    x = synth.synth_image_batch(1000, 32, 544, 544)
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80).eval()                   # default precision: what build(config['model'], ...) runs
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    post = _hip_post((544, 544), dev)
    pick = [3, 17, 30]
    with torch.no_grad():
        out = net(x.to(dev))
        assert out.precision == "f32_split" and out.flags() == 0
        res = post(out)
    assert len(res) == 32 and all(r["bbox"].shape[0] == 100 for r in res)         # dense heads: 100 detections per image
    heads = [(b[pick].cpu(), o[pick].cpu()) for b, o in out]
    pc = post_cfg((544, 544))
    oracle_post = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80,
                                      conf_thresh=pc["conf_thresh"])
    want_same = oracle_post(heads)
    ties = 0
    for i, b in enumerate(pick):
        r = res[b]
        gb, gc, gm = _sorted_within_ties(r["bbox"].cpu().numpy(), r["cls"].cpu().numpy(), r["mask"].cpu().numpy())
        wb, wc, wm = _sorted_within_ties(want_same[i]["bbox"].numpy(), want_same[i]["cls"].numpy(), want_same[i]["mask"].numpy())
        ties += wb.shape[0] - np.unique(wb[:, 4]).size
        assert np.array_equal(gc, wc), b
        assert np.array_equal(gb[:, [0, 1, 4]], wb[:, [0, 1, 4]]) and _ulps(gb[:, 2:4], wb[:, 2:4]).max() <= 2, b
        assert all(_mask_iou(gm[k], wm[k]) >= 1 - 1e-4 for k in range(gm.shape[0])), b
    print("bench workload: %d detections of the 3 checked images sit in exact-tie groups" % ties)
    ref = R.forward(sd, x[pick])
    for (gb, go), (rb, ro) in zip(heads, ref):
        assert _rel_err(gb, rb) < REL_TOL and _rel_err(go, ro) < REL_TOL
    want_e2e = oracle_post(ref)
    for i, b in enumerate(pick):
        _check_detections_composed(res[b], want_e2e[i]["bbox"].numpy(), want_e2e[i]["cls"].numpy(), want_e2e[i]["mask"].numpy(),
                                   ("bench workload end to end", b), margin_ctx=_margin_ctx(oracle_post, heads, i))


@pytest.mark.parametrize("precision", ["f32", "f32_split"])
def test_backbone_features_bs8_match_oracle(dev, precision):
    """BASELINE configs[1]: DarkNet-53 only, random weights, bs=8 at 544x544 -- x4 / x8 / x16 / x32 as the HIP kernels
    left them in the workspace vs the CPU oracle's backbone, <= 1e-4 of each tensor's scale; with fp32 operands and in the
    plugin's / the bench's default precision (split operands)."""
    sd = synth.synth_state_dict(8, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(25, 8, 544, 544)
    net = _hip_model(sd, dev, precision)
    with pytest.raises(omlib.OrienMaskHipError):
        net(torch.zeros(1, 3, 64, 64, device=dev)); net.layer_output("backbone.conv3.2.conv.1", (1, 3, 64, 64))
    net.keep_activations(True)          # activations share memory by live range unless asked to stay
    with torch.no_grad():
        net(x.to(dev))
    torch.cuda.synchronize()
    x32, x16, x8, x4 = _oracle_once(("backbone", 8, 25, 8, 544, 544), lambda: R.backbone(sd, x))
    for name, want in (("backbone.conv3.2.conv.1", x4), ("backbone.conv4.8.conv.1", x8), ("backbone.conv5.8.conv.1", x16),
                       ("backbone.conv6.4.conv.1", x32)):
        got = net.layer_output(name, x.shape).cpu()
        assert got.shape == want.shape
        assert _rel_err(got, want) < REL_TOL, name


@pytest.mark.parametrize("prec", ["f32", "f16"])
def test_forward_on_side_streams_is_bit_identical(dev, prec):
    """set_streams(2): two sub-batches on two HIP streams == one launch, bit for bit; odd batches fall back to one launch."""
    sd = synth.synth_state_dict(7, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(24, 4, 544, 544).to(dev)
    net = _hip_model(sd, dev).set_precision(prec)
    with torch.no_grad():
        want = [(a.clone(), b.clone()) for a, b in net(x)]
        net.set_streams(2)
        for xb, wb in ((x, want), (x[:3], [(a[:3], b[:3]) for a, b in want])):
            got = net(xb)
            torch.cuda.synchronize()
            for (ga, gb), (wa, wb_) in zip(got, wb):
                assert torch.equal(ga, wa) and torch.equal(gb, wb_)


@pytest.mark.parametrize("prec,depth,bs0", [("f32", 2, 2), ("f16", 3, 2), ("f32_split", 2, 6)])
def test_in_flight_pipeline_matches_eager(dev, prec, depth, bs0):
    """InFlightPipeline (whole batches on alternating HIP streams, own workspaces per slot) returns, in submission order,
    exactly what the one-batch-at-a-time loop returns; different batch sizes pass through the same slots.  (Split operands with
    batches of 6 / 7: the F(2x4) kernels, incl. their stream-K form with its inter-workgroup hand-off, beside another batch.)"""
    from orienmask_amd.pipeline import InFlightPipeline
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net, net_ref = _hip_model(sd, dev).set_precision(prec), _hip_model(sd, dev).set_precision(prec)
    post, post_ref = _hip_post((544, 544), dev), _hip_post((544, 544), dev)
    xs = [synth.synth_image_batch(900 + i, bs0 + (i % 2), 544, 544).to(dev) for i in range(7)]
    pipe = InFlightPipeline(net, post, depth=depth)
    got = [[{k: v.clone() for k, v in d.items()} for d in dets] for dets in pipe.map(xs)]
    assert len(pipe) == 0 and len(got) == len(xs)
    for x, g_list in zip(xs, got):
        with torch.no_grad():
            want = post_ref(net_ref(x))
        assert len(g_list) == len(want) == x.shape[0]
        for g, w in zip(g_list, want):
            assert g["bbox"].shape[0] > 0
            assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"])
    # submit / result by hand: at most `depth` pending, results in order
    for i in range(depth):
        pipe.submit(xs[i])
    with pytest.raises(RuntimeError):
        pipe.submit(xs[0])
    for i in range(depth):
        for g, w in zip(pipe.result(), got[i]):
            assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["mask"], w["mask"])
    with pytest.raises(RuntimeError):
        pipe.result()
    with pytest.raises(omlib.OrienMaskHipError):
        pipe.submit(xs[0].cpu())


@pytest.mark.parametrize("prec,other", [("f32", "f16"), ("f32", "f32"), ("f16", "f16"), ("f32_split", "f32_split")])
def test_postprocess_and_forward_are_stable_beside_other_streams(dev, prec, other):
    """Kernels of two HIP streams share compute units (InFlightPipeline relies on it).  Results must not depend on what the
    other stream runs: the postprocess of a fixed prediction and a forward of a fixed batch are repeated while a second
    model instance runs forwards on another stream, and compared bit for bit with the launch that ran alone.
    (The mask kernel's floating-point-compare predicate failed exactly this beside fp16 convolutions: post.hip inside_bit,
    tools/hazard_probe.)"""
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net = _hip_model(sd, dev).set_precision(prec)
    busy = _hip_model(sd, dev).set_precision(other)
    post = _hip_post((544, 544), dev)
    split = prec == "f32_split"           # batches of 6: the F(2x4) split-operand kernels (wide-K fp16 matrix instructions everywhere)
    x = synth.synth_image_batch(900, 6 if split else 2, 544, 544).to(dev)
    y = synth.synth_image_batch(901, 6 if split else 1, 544, 544).to(dev)
    s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.no_grad():
        pred = net(x)
        want_heads = [(a.clone(), b.clone()) for a, b in pred]
        want = [{k: v.clone() for k, v in d.items()} for d in post(pred)]
        want_keep = [k.clone() for k in post.last_keep]
        busy(y)
        torch.cuda.synchronize()
        for it in range(8):
            with torch.cuda.stream(s1):
                for _ in range(6 if other == "f16" else (3 if split else 2)):
                    busy(y)
            with torch.cuda.stream(s0):
                outs = post.launch(pred)
                heads = net(x)
            torch.cuda.synchronize()
            got = post.collect(outs)
            for g, w, gk, wk in zip(got, want, post.last_keep, want_keep):
                assert torch.equal(gk, wk), it
                assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]), it
                assert torch.equal(g["mask"], w["mask"]), (it, int((g["mask"] != w["mask"]).sum()))
            for (ga, gb), (wa, wb) in zip(heads, want_heads):
                assert torch.equal(ga, wa) and torch.equal(gb, wb), it


@pytest.mark.parametrize("prec,latency,B", [("f32_split", True, 1), ("f32_split", False, 3), ("f32", False, 2), ("f32_split", True, 2)])
def test_launch_step_same_bits_as_two_calls(dev, prec, latency, B):
    """eval.launch_step / om_model_attach_postprocess: the postprocess attached to the forward -- decode + select on the library's
    second stream as soon as the box heads are launched, the mask kernel behind the last layer -- against postprocess(model(x)):
    every output bit for bit, eagerly back to back without a synchronisation in between (the next forward must not run into the
    previous step's second stream: the join), with the attachment gone afterwards, and through a captured hipGraph."""
    from orienmask_amd.graph import GraphedPipeline
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net = _hip_model(sd, dev, prec)
    if latency:
        net.set_latency_mode(True)
    post = _hip_post((544, 544), dev)
    xs = [synth.synth_image_batch(930 + i, B, 544, 544).to(dev) for i in range(2)]

    def same(got, want, tag):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"]), tag

    with torch.no_grad():
        want, want_keep, want_heads = [], [], []
        for x in xs:
            out = net(x)
            want_heads.append([(a.clone(), b.clone()) for a, b in out])
            want.append([{k: v.clone() for k, v in d.items()} for d in post(out)])
            want_keep.append([k.clone() for k in post.last_keep])
        assert sum(len(d["cls"]) for d in want[0]) > 0
        for rep in range(3):
            outs = [post.launch_step(net, xs[k & 1]) for k in range(4)]      # back to back, nothing synchronises in between
            for k, o in enumerate(outs):
                same(post.collect(o), want[k & 1], (rep, k))
                for a, b in zip(post.last_keep, want_keep[k & 1]):
                    assert torch.equal(a, b)
                for (ga, gb), (wa, wb) in zip(o[6], want_heads[k & 1]):      # the forward's own outputs
                    assert torch.equal(ga, wa) and torch.equal(gb, wb)
        # detached again: a plain forward launches no postprocess (the attached buffers are untouched by it)
        o = post.launch_step(net, xs[0])
        torch.cuda.synchronize()
        before = o[0].clone()
        o[0].fill_(-7.0)
        net(xs[1])
        torch.cuda.synchronize()
        assert bool((o[0] == -7.0).all()) and before.numel() > 0
        gp = GraphedPipeline(net, post, xs[0])
        for k in range(5):
            same(gp(xs[k & 1]), want[k & 1], ("graph", k))


@pytest.mark.parametrize("other", ["f32_split", "f16"])
def test_latency_mode_split_k_is_stable_beside_other_streams(dev, other):
    """The split-K parts of the latency mode hand their accumulators to the last arrival through memory WITHOUT fences
    (write-through stores, drained waves, one agent-scope atomic, sc1 loads: conv_igemm_split.hip).  That protocol has to hold
    under uneven load and with the partial-tile area reused by other data between launches: single-image forwards of two
    different images, alternating, while another model instance keeps the chip busy on a second stream -- every head bit for bit
    equal to the forward of the same image that ran alone."""
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    net = _hip_model(sd, dev, "f32_split").set_latency_mode(True)
    busy = _hip_model(sd, dev).set_precision(other)
    xs = [synth.synth_image_batch(910 + i, 1, 544, 544).to(dev) for i in range(2)]
    y = synth.synth_image_batch(920, 6 if other == "f32_split" else 4, 544, 544).to(dev)
    s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.no_grad():
        kernels = dict(net.layer_kernels(1, 544, 544))
        assert kernels["backbone.conv6.1.conv.1"].startswith("conv_igemm_split_kernel<64,64"), kernels["backbone.conv6.1.conv.1"]
        want = []
        for x in xs:
            out = net(x)
            assert out.flags() == 0
            want.append([(a.clone(), b.clone()) for a, b in out])
        assert not torch.equal(want[0][0][0], want[1][0][0])
        busy(y)
        torch.cuda.synchronize()
        for it in range(10):
            with torch.cuda.stream(s1):
                for _ in range(4 if other == "f16" else 2):
                    busy(y)
            got = []
            with torch.cuda.stream(s0):
                for k in range(6):
                    got.append((k & 1, [(a.clone(), b.clone()) for a, b in net(xs[k & 1])]))
            torch.cuda.synchronize()
            for which, heads in got:
                for (ga, gb), (wa, wb) in zip(heads, want[which]):
                    assert torch.equal(ga, wa) and torch.equal(gb, wb), (it, which)


def test_forward_f16_non_plus_model_matches_oracle(dev):
    """The OrienMaskYOLO graph (variant 1) through the fp16 path."""
    from orienmask_amd.model import OrienMaskYOLO
    sd = synth.synth_state_dict(6, obj_bias=-16.0, head_gain=4.0, model="OrienMaskYOLO")
    x = synth.synth_image_batch(23, 1, 544, 544)
    net = OrienMaskYOLO(3, 80).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).set_precision("f16")
    with torch.no_grad():
        out = net(x.to(dev))
    want = R.forward_f16(sd, x, model="OrienMaskYOLO")
    for (gb, go), (wb, wo) in zip(out, want):
        assert _rel_err(gb.cpu(), wb) < F16_FWD_TOL and _rel_err(go.cpu(), wo) < F16_FWD_TOL


def test_forward_f16_is_batch_invariant_and_switchable(dev):
    """Same image alone or inside a batch -> bit-identical heads; switching the precision back gives the f32 results."""
    sd = synth.synth_state_dict(5, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(22, 3, 544, 544).to(dev)
    net = _hip_model(sd, dev)
    with torch.no_grad():
        f32_before = [(a.clone(), b.clone()) for a, b in net(x[:1])]
        net.set_precision("f16")
        full = [(a.clone(), b.clone()) for a, b in net(x)]
        one = net(x[1:2])
        for (fb, fo), (ob, oo) in zip(full, one):
            assert torch.equal(fb[1:2], ob) and torch.equal(fo[1:2], oo)
        net.set_precision("f32")
        for (a, b), (c, d) in zip(net(x[:1]), f32_before):
            assert torch.equal(a, c) and torch.equal(b, d)


def test_forward_f16_bs64_matches_oracle(dev):
    """BASELINE configs[4] at its own per-GPU batch: 64 x 544x544 through om_forward_f16 in one call.  Two images of the batch
    (first, last) against oracle.forward_f16 -- the definition of this configuration's arithmetic; parity with the reference is
    unpinned by construction, the reference has no reduced-precision path -- and the whole batch bit-identical to a batch of 4
    holding the same images (other tile shapes and grids)."""
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(64, 64, 544, 544)
    net = _hip_model(sd, dev).set_precision("f16")
    pick = [0, 63]
    with torch.no_grad():
        out = net(x.to(dev))
        heads = [(b[pick].clone(), o[pick].clone()) for b, o in out]
        assert all(torch.isfinite(b).all() and torch.isfinite(o).all() for b, o in out)
        small = net(x[pick + [7, 8]].to(dev))
    for (hb, ho), (sb, so) in zip(heads, small):
        assert torch.equal(hb, sb[:2]) and torch.equal(ho, so[:2])
    want = R.forward_f16(sd, x[pick])
    for (gb, go), (wb, wo) in zip(heads, want):
        assert _rel_err(gb.cpu(), wb) < F16_FWD_TOL and _rel_err(go.cpu(), wo) < F16_FWD_TOL


@pytest.mark.parametrize("xseed", [31, 32, 33, 34, 35])
def test_end_to_end_f16_agrees_with_f32(dev, xseed):
    """Detections of the fp16 configuration against the fp32 path: the detections in the upper half of each image's
    score range pair up (same class, box IoU > 0.9, mask IoU > 0.9) for at least 90 % of them."""
    sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(xseed, 2, 544, 544).to(dev)
    net = _hip_model(sd, dev)
    post = _hip_post((544, 544), dev)
    with torch.no_grad():
        ref = [{k: v.clone() for k, v in d.items()} for d in post(net(x))]
        assert all(r["bbox"].shape[0] > 0 for r in ref)
        net.set_precision("f16")
        got = post(net(x))
    for r, g in zip(ref, got):
        conf = r["bbox"][:, 4] >= 0.5 * r["bbox"][:, 4].max()        # the upper half of the score range
        matched = 0
        for i in torch.nonzero(conf).flatten().tolist():
            same = torch.nonzero(g["cls"] == r["cls"][i]).flatten()
            if same.numel() == 0:
                continue
            bi = r["bbox"][i, :4]; bj = g["bbox"][same, :4]
            lt = torch.maximum(bi[:2] - bi[2:] / 2, bj[:, :2] - bj[:, 2:] / 2)
            rb = torch.minimum(bi[:2] + bi[2:] / 2, bj[:, :2] + bj[:, 2:] / 2)
            inter = (rb - lt).clamp(min=0).prod(1)
            iou = inter / (bi[2:].prod() + bj[:, 2:].prod(1) - inter)
            j = int(iou.argmax())
            mi, mj = r["mask"][i], g["mask"][same[j]]
            miou = (mi & mj).sum().item() / max((mi | mj).sum().item(), 1)
            matched += int(iou[j] > 0.9 and miou > 0.9)
        assert matched >= 0.9 * int(conf.sum()), (matched, int(conf.sum()))


# ------------------------------------------------------------------------------------------------
# the range guard of the default precision (include/orienmask_hip.h: OM_STATUS_SPLIT_RANGE)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gain", [3.0e3, 1.0e5])
def test_forward_range_guard_falls_back_to_f32(dev, gain):
    """precision 'f32_split' is the plugin's default, so its range condition is guarded on the device: an input scaled until
    activations leave fp16's range (gain 1e5: already in front of the first stride-2 layer; 3e3: only the TRANSFORMED input of
    a stride-1 3x3 layer, up to 20x the activation, does) sets OM_STATUS_SPLIT_RANGE through om_forward, and what the caller
    gets -- Prediction.check(), or simply postprocess(model(x)) -- is the fp32-operand result, bit for bit.  An ordinary input
    leaves the status word clear."""
    import warnings
    from orienmask_amd import model as om_model
    assert om_model.OrienMaskYOLOFPNPlus(3, 80).precision == "f32_split"        # the default IS the guarded mode
    sd = synth.synth_state_dict(12, obj_bias=-3.0, head_gain=0.7)
    x = synth.synth_image_batch(77, 2, 160, 128)
    xs = (x * gain).to(dev)
    net = _hip_model(sd, dev, "f32_split")
    ref = _hip_model(sd, dev, "f32")
    post = _hip_post((160, 128), dev)
    with torch.no_grad():
        clean = net(x.to(dev))
        assert clean.precision == "f32_split" and clean.flags() == 0 and clean.check() is clean
        pred = net(xs)
        assert pred.flags() == omlib.OM_STATUS_SPLIT_RANGE
        assert not all(bool(torch.isfinite(b).all()) for b, _ in pred)          # the split forward's heads are NaN, and say so
        want = ref(xs)
        assert want.flags() == 0 and all(bool(torch.isfinite(b).all() and torch.isfinite(o).all()) for b, o in want)
        om_model._warned_range = False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            fixed = pred.check()
        assert any("re-run with fp32 operands" in str(m.message) for m in w)
        assert fixed.precision == "f32"
        for (gb, go), (wb, wo) in zip(fixed, want):
            assert torch.equal(gb, wb) and torch.equal(go, wo)
        dets, dets_want = post(pred), post(want)                                # the postprocess resolves the status itself
    for d, dw in zip(dets, dets_want):
        assert all(torch.equal(d[k], dw[k]) for k in ("bbox", "cls", "mask"))
    # ... and inside the in-flight pipeline, between two ordinary batches
    from orienmask_amd.pipeline import InFlightPipeline
    pipe = InFlightPipeline(net, post, depth=2)
    with torch.no_grad():
        out = list(pipe.map([x.to(dev), xs, x.to(dev)]))
        eager_clean = post(net(x.to(dev)))
    for d, dw in zip(out[1], dets_want):
        assert all(torch.equal(d[k], dw[k]) for k in ("bbox", "cls", "mask"))
    for batch in (out[0], out[2]):
        for d, dw in zip(batch, eager_clean):
            assert all(torch.equal(d[k], dw[k]) for k in ("bbox", "cls", "mask"))


def test_rccl_broadcast_of_every_blob_on_one_gpu(dev):
    """The RCCL call path of the N > 1 bench (orienmask_amd/dist.py: broadcast_packed_weights) with world size 1 on the nccl
    backend: the raw fp32 state_dict (~255 MB) travels through dist.broadcast as ONE blob, the rank packs it on its
    own device -- the fp32 blob, the split blob of the default precision, the fp16 rows -- and the bound model computes the same
    heads as one that packed its own weights; a rank that did NOT have the weights ends up with them in its module."""
    import socket
    import torch.distributed as dist
    from orienmask_amd.dist import broadcast_packed_weights
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = synth.synth_state_dict(14, obj_bias=-16.0, head_gain=4.0)
        x = synth.synth_image_batch(15, 2, 96, 128).to(dev)
        for prec in ("f32_split", "f16"):
            net = OrienMaskYOLOFPNPlus(3, 80, precision=prec).eval()
            net.load_state_dict(sd, strict=True)
            stats = {}
            broadcast_packed_weights(net, dev, src=0, stats=stats, verify=True)
            assert stats["blobs"] == 1 and 2.54e8 < stats["bytes"] < 2.56e8 and stats["blobs_identical_across_ranks"] is True
            assert net._packed is not None and (net._packed_split if prec == "f32_split" else net._packed16) is not None
            own = _hip_model(sd, dev, prec)
            with torch.no_grad():
                a, b = net(x), own(x)
            for (ab, ao), (bb, bo) in zip(a, b):
                assert torch.equal(ab, bb) and torch.equal(ao, bo)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_forward_f16_after_set_precision_split_c_api(dev):
    """a C-API caller may leave the model in precision mode 1 (split operands) and then call
    om_forward_f16.  The fp16 forward must not take mode 1's fused conv1 + conv2.0 kernel (fp32 stores into a buffer sized for
    2-byte elements, split weights that need not be loaded): same heads, bit for bit, as with mode 0."""
    from orienmask_amd import lib as omlib
    sd = synth.synth_state_dict(5, obj_bias=-16.0, head_gain=4.0)
    x = synth.synth_image_batch(23, 2, 96, 128).to(dev)
    net = _hip_model(sd, dev, "f16")
    with torch.no_grad():
        want = [(a.clone(), b.clone()) for a, b in net(x)]           # the wrapper sets mode 0 before an fp16 forward
    L = omlib.load()
    h = net._handle
    B, _, H, W = x.shape
    heads = [torch.empty((B, H // s, W // s, 256), dtype=torch.float32, device=dev) for s in (32, 16, 8)]
    oriens = torch.empty((B, 18, H // 4, W // 4), dtype=torch.float32, device=dev)
    omlib.check(L.om_model_set_precision(h, 1), "om_model_set_precision")     # no split weights were ever loaded
    nbytes = (L.om_forward_f16_workspace_bytes(h, B, H, W) + 255) // 256 * 256
    ws = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=dev)
    ws[nbytes:] = 0x5A                                                          # guard behind the workspace
    omlib.check(L.om_forward_f16(h, _p(x), B, H, W, _p(heads[0]), _p(heads[1]), _p(heads[2]), _p(oriens), _p(ws), nbytes,
                                 omlib.current_stream_ptr(dev)), "om_forward_f16")
    torch.cuda.synchronize()
    omlib.check(L.om_model_set_precision(h, 0), "om_model_set_precision")
    assert bool((ws[nbytes:] == 0x5A).all()), "the forward wrote behind its workspace"
    got_o = torch.split(oriens, 6, dim=1)
    for i, (wb, wo) in enumerate(want):
        assert torch.equal(heads[i][..., :255].permute(0, 3, 1, 2), wb) and torch.equal(got_o[i], wo)


def test_coco_strings_packed_on_the_device(dev):
    """pycocotools' rleToString runs as the RLE kernel's last phase (om_recover_masks_rle_strings), every
    mask of a batch appends to one byte buffer.  Byte-identical to oracle/rle_ref.c on the reference-generated resized masks of
    tests/golden/coco_format.npz -- all five cases as ONE batch -- and again with first-guess buffers so small that every mask
    takes an overflow path (too many runs for the run buffer; strings beyond the byte buffer)."""
    from orienmask_amd.coco_format import COCOFormatter
    from test_oracle_golden import _coco_cases
    infos, dets, want = [], [], []
    for i, (name, info, masks, bbox, xywh, seg) in enumerate(_coco_cases()):
        K = masks.shape[0]
        b5 = torch.cat([torch.from_numpy(bbox)[:, :4], torch.linspace(0.9, 0.1, K).reshape(K, 1)], dim=1).float()
        infos.append(dict(info, id=100 + i))
        dets.append(dict(bbox=b5.to(dev), cls=torch.arange(K, dtype=torch.int64, device=dev) % 80, mask=torch.from_numpy(masks).to(dev)))
        want += [(100 + i, [info["height"], info["width"]], R.rle_to_string(R.rle_counts(seg[k])), xywh[k].tolist()) for k in range(K)]
    # an image without detections in the middle of the batch contributes nothing
    infos.insert(2, dict(id=7, height=50, width=60))
    dets.insert(2, dict(bbox=torch.zeros((0, 5), device=dev), cls=torch.zeros((0,), dtype=torch.int64, device=dev),
                        mask=torch.zeros((0, 96, 128), dtype=torch.bool, device=dev)))
    for max_runs, per_mask, worst in ((8192, 4096, None), (4, 4096, None), (8192, 3, None), (4, 4096, 1)):
        fmt = COCOFormatter(list(range(1, 81)), with_mask=True)
        fmt.MAX_RUNS, fmt.BYTES_PER_MASK = max_runs, per_mask
        if worst is not None:
            fmt.WORST_CASE_BYTES = worst         # every overflowing mask a worst-case launch of its own (the chunked fallback)
        res = fmt.to_coco_format(infos, dets)
        res = fmt.to_coco_format(infos, dets)    # ... and again from the formatter's cached scratch buffers
        assert len(res["segm"]) == len(res["bbox"]) == len(want)
        for s, b, (iid, size, counts, box) in zip(res["segm"], res["bbox"], want):
            assert s["image_id"] == b["image_id"] == iid and s["segmentation"]["size"] == size and b["bbox"] == box
            assert s["segmentation"]["counts"] == counts, (max_runs, per_mask, iid)
    # an output too large for the LDS bitmap (1400 x 1000 -> 44 words x 1000 columns) takes the old kernel + the string kernel
    big = dict(id=1, height=1400, width=1000)
    m = torch.zeros((2, 96, 128), dtype=torch.bool)
    m[0, 20:60, 30:90] = True
    m[1, ::3, ::5] = True
    res = COCOFormatter(list(range(1, 81))).to_coco_format([big], [dict(bbox=torch.rand(2, 5).to(dev), cls=torch.zeros(2, dtype=torch.int64, device=dev), mask=m.to(dev))])
    seg = R.recover_shape_segm(m, big).numpy()
    assert [s["segmentation"]["counts"] for s in res["segm"]] == [R.rle_to_string(R.rle_counts(seg[k])) for k in range(2)]


# ------------------------------------------------------------------------------------------------
# the two generalities of the reference's postprocess the fused path used to refuse
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("regime,seed", [("mixed", 201), ("sparse_many", 202), ("clustered", 203), ("empty", 204), ("dense", 205)])
def test_postprocess_foreign_nms_callable(dev, regime, seed):
    """nms_func may be ANY callable (dets[n,5], cls[n]) -> (dets[keep], cls[keep], keep), as in the reference
    (/root/reference/eval/orienmask_yolo_postprocess.py:9-11,146-154).  (a) a plain function that happens to do batched_nms: the
    three-stage path (om_postprocess_candidates -> callable -> om_postprocess_masks) equals the fused kernel bit for bit;
    (b) class-agnostic NMS at another threshold, keep in another order: equals the oracle's pieces composed the same way."""
    from orienmask_amd import eval as om_eval
    size = (160, 192)
    pc = post_cfg(size)
    heads = synth.synth_heads(seed, 3, pc["grid_size"], regime=regime)
    dheads = tuple((b.to(dev), o.to(dev)) for b, o in heads)
    fused = _hip_post(size, dev)
    want = fused(dheads)
    want_keep = [k.clone() for k in fused.last_keep]
    calls = []

    def plain(dets, cls):
        calls.append(int(dets.shape[0]))
        assert dets.is_cuda and cls.dtype == torch.long
        return om_eval.batched_nms(dets, cls, threshold=0.5)

    post = _hip_post(size, dev, nms_func=plain)
    got = post(dheads)
    assert post.nms_thresh is None and (len(calls) > 0 or regime == "empty")
    for b, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"]), (regime, b)
        assert torch.equal(post.last_keep[b], want_keep[b])
    # (b) class-agnostic, reversed keep order: against the oracle's candidates + its nms + its mask rule
    def agnostic(dets, cls):
        keep = om_eval._nms_keep(dets, 0.3, "cpu").flip(0)
        return dets[keep], cls[keep], keep

    got = _hip_post(size, dev, nms_func=agnostic)(dheads)
    oracle = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 80, conf_thresh=pc["conf_thresh"])
    cpu_heads = [(b.float(), o.float()) for b, o in heads]
    for b in range(3):
        coord, score, cls, aidx, sel = oracle.candidates(cpu_heads, b)
        dets = torch.cat([coord, score.unsqueeze(-1)], 1)
        keep = R.nms_cpu(dets, 0.3).flip(0)
        if keep.numel() > 100:
            keep = keep[dets[keep][:, -1].topk(100)[1]]
        field = oracle.orien_field(cpu_heads, b)
        a = aidx[keep]
        gsx, gsy = oracle.grid_sizes[a, 0], oracle.grid_sizes[a, 1]
        d = dets[keep]
        masks = ((torch.abs(field[a, 0] - (gsx * d[:, 0]).view(-1, 1, 1)) < 0.3 * d[:, 2].view(-1, 1, 1) * gsx.view(-1, 1, 1)) &
                 (torch.abs(field[a, 1] - (gsy * d[:, 1]).view(-1, 1, 1)) < 0.3 * d[:, 3].view(-1, 1, 1) * gsy.view(-1, 1, 1)))
        _check_detections(got[b], d.numpy(), cls[keep].numpy(), masks.numpy(), (regime, "agnostic", b), exact_decode=True)


@pytest.mark.parametrize("masks,slices", [([[6, 7, 8], [3, 4, 5]], (3, 3)), ([[6, 7, 8], [3, 4, 5], [0, 1]], (3, 3, 2)),
                                          ([[4, 8]], (2,)), ([[6], [3, 5], [0, 1, 2]], (1, 2, 3))])
@pytest.mark.parametrize("regime", ["mixed", "sparse_many"])
def test_postprocess_other_scale_and_anchor_counts(dev, masks, slices, regime):
    """The reference builds its tables for any number of scales and anchors per scale (postprocess.py:13-36).  1..3 scales of
    1..3 anchors each against the oracle, which is generic like the reference: indices, classes and keep exact, boxes and masks
    as for the standard configuration; heads both as plain NCHW tensors and in one shared orientation buffer."""
    from orienmask_amd.eval import OrienMaskYOLOPostProcess
    size = (160, 192)
    pc = post_cfg(size)
    full = synth.synth_heads(77, 2, pc["grid_size"], regime=regime)
    heads = [(full[i][0][:, :slices[i] * 85].contiguous(), full[i][1][:, :slices[i] * 2].contiguous()) for i in range(len(masks))]
    grids = pc["grid_size"][:len(masks)]
    oracle = R.PostProcessOracle(grids, pc["image_size"], pc["anchors"], masks, 80, conf_thresh=pc["conf_thresh"])
    want = oracle(heads)
    post = OrienMaskYOLOPostProcess(grids, pc["image_size"], pc["anchors"], masks, 80, conf_thresh=pc["conf_thresh"], device=dev)
    shared = torch.cat([o for _, o in heads], 1).contiguous().to(dev)
    views = torch.split(shared, [s * 2 for s in slices], dim=1)
    for layout in ("plain", "shared"):
        pred = tuple((b.to(dev), o.to(dev) if layout == "plain" else views[i]) for i, (b, o) in enumerate(heads))
        got = post(pred)
        for b, (r, w) in enumerate(zip(got, want)):
            assert torch.equal(r["cls"].cpu(), w["cls"]) and torch.equal(post.last_keep[b].cpu().long(), w["keep"]), (masks, layout, b)
            _check_detections(r, w["bbox"].numpy(), w["cls"].numpy(), w["mask"].numpy(), (masks, layout, b), exact_decode=True)


@pytest.mark.parametrize("C,regime", [(1, "mixed"), (20, "sparse_many"), (251, "mixed"), (252, "mixed"), (300, "dense"), (1203, "mixed"),
                                      (2047, "sparse")])
def test_postprocess_class_counts(dev, C, regime):
    """The reference takes any num_classes (postprocess.py:13-36).  The decode kernel unrolls a thread's visits per sweep: ten
    cover C <= 251 (the COCO configuration's 80 among them), a second instantiation the counts up to 2047 (LVIS: 1203); both
    against the oracle -- indices, classes and keep exact, boxes and masks as for the standard configuration -- and one class more
    than the library holds is refused loudly."""
    from orienmask_amd.eval import OrienMaskYOLOPostProcess
    size = (96, 128)
    pc = post_cfg(size)
    heads = synth.synth_heads(300 + C, 2, pc["grid_size"], num_classes=C, regime=regime)
    oracle = R.PostProcessOracle(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], C, conf_thresh=pc["conf_thresh"])
    want = oracle(heads)
    post = OrienMaskYOLOPostProcess(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], C,
                                    conf_thresh=pc["conf_thresh"], device=dev)
    got = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
    assert sum(len(w["cls"]) for w in want) > 0
    for b, (r, w) in enumerate(zip(got, want)):
        assert torch.equal(r["cls"].cpu(), w["cls"]) and torch.equal(post.last_keep[b].cpu().long(), w["keep"]), (C, b)
        _check_detections(r, w["bbox"].numpy(), w["cls"].numpy(), w["mask"].numpy(), (C, b), exact_decode=True)
    if C == 2047:
        too_many = OrienMaskYOLOPostProcess(pc["grid_size"], pc["image_size"], pc["anchors"], pc["anchor_mask"], 4100,
                                            conf_thresh=pc["conf_thresh"], device=dev)
        big = synth.synth_heads(1, 1, pc["grid_size"], num_classes=4100, regime="sparse")
        with pytest.raises(omlib.OrienMaskHipError, match="num_classes"):
            too_many(tuple((b.to(dev), o.to(dev)) for b, o in big))


@pytest.mark.parametrize("fname", ["post_p544_ties_iou_b1.npz", "post_p544_ties_cut_b2.npz", "post_p544_ties_thresh_b2.npz",
                                   "post_p544_dense_b1.npz"])
def test_near_tie_fixtures_beside_fp16_matrix_neighbour(dev, fname):
    """the comparisons that decide INDICES (confidence against conf_thresh, IoU against the NMS
    threshold, the order of the sort, the radix select's key tests) are evaluated on bit patterns in VGPRs since round 4
    (csrc/post.hip: f32_gt_bit / f32_ge_bit / lt_u32_bit ...), because a dense run of VALU compares into SGPR pairs returned stale
    lane masks on this chip while a co-resident wave issued wide-K fp16 matrix instructions (tools/hazard_probe).  The
    reference-generated near-tie fixtures -- scores 1-3 ulps apart across the nms_pre cut, confidences stepping through
    conf_thresh two ulps at a time, IoUs stepping through 0.5 a fraction of an ulp at a time, and the dense fixture that takes
    the radix-select path -- must come out identical to the reference's own results on every one of 12 repetitions while
    another stream runs this library's fp16 convolutions (v_mfma_f32_32x32x16_f16 on every CU) without a pause."""
    g = np.load(os.path.join(GOLDEN, fname))
    size = tuple(int(v) for v in g["size"]); batch = int(g["batch"])
    pc = post_cfg(size)
    heads = synth.synth_heads(int(g["seed"]), batch, pc["grid_size"], regime=str(g["regime"]))
    dheads = _to_model_layout(heads, dev)
    post = _hip_post(size, dev)
    busy = _hip_model(synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0), dev).set_precision("f16")
    y = synth.synth_image_batch(901, 2, 544, 544).to(dev)
    s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.no_grad():
        busy(y)
        alone = post(dheads)
        alone_keep = [k.clone() for k in post.last_keep]
        torch.cuda.synchronize()
        for it in range(12):
            with torch.cuda.stream(s1):
                for _ in range(4):
                    busy(y)
            with torch.cuda.stream(s0):
                outs = post.launch(dheads)
            torch.cuda.synchronize()
            res = post.collect(outs)
            for b, r in enumerate(res):
                _check_detections(r, g["bbox%d" % b], g["cls%d" % b], unpack_masks(g["mask%d" % b], g["maskshape%d" % b]),
                                  (fname, "beside fp16", it, b), exact_decode=True)
                assert torch.equal(r["bbox"], alone[b]["bbox"]) and torch.equal(r["cls"], alone[b]["cls"]) and \
                    torch.equal(r["mask"], alone[b]["mask"]) and torch.equal(post.last_keep[b], alone_keep[b]), (fname, it, b)


@pytest.mark.parametrize("fname", ["fwd_f544_b1.npz", "fwd_f160x128_b1.npz", "fwd_stress_f160x128_b1.npz"])
def test_latency_mode_matches_reference_golden(dev, fname):
    """model.set_latency_mode (om_model_set_latency_cells): batches of a few images run their stride-1
    3x3 layers as direct split-operand convolutions in the implicit GEMM instead of the fused F(4,3) kernel.  Same bar as every
    other forward: heads within 1e-4 of the REFERENCE's tensors (tests/golden/fwd_*.npz), detections of the composed path as in
    test_forward_matches_reference_golden; against the default mode ~1e-6 of scale, and -- the point of the mode being opt-in --
    not bit-identical to it; above the switch (a batch of 6 images) the mode changes nothing, bit for bit."""
    g = np.load(os.path.join(GOLDEN, fname))
    size = tuple(int(v) for v in g["size"])
    sd, x = fixture_weights_and_input(g)
    net = _hip_model(sd, dev, "f32_split")
    xd = x.to(dev)
    with torch.no_grad():
        base = [(a.clone(), b.clone()) for a, b in net(xd)]
        net.set_latency_mode(True)
        kernels = dict(net.layer_kernels(xd.shape[0], size[0], size[1]))
        assert kernels["backbone.conv4.1.conv.1"].startswith("conv_igemm_split_kernel"), kernels["backbone.conv4.1.conv.1"]
        out = net(xd)
        assert out.flags() == 0
        _check_forward_fixture(out, g, fname, "f32_split latency mode", dev, size, int(g["batch"]))
        for i, (gb, go) in enumerate(out):
            assert _rel_err(gb, base[i][0]) < 2e-5 and _rel_err(go, base[i][1]) < 2e-5
        assert not all(torch.equal(a, c) for (a, _), (c, _) in zip(out, base)), "the latency mode did not change the arithmetic"
        again = net(xd)
        assert all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(out, again))      # run-to-run identical
        # a batch above the switch is untouched by the mode
        big = torch.cat([xd] * 6) if size[0] * size[1] >= 544 * 544 else None
        if big is not None:
            net.set_latency_mode(True, cells=1200)      # six images: 1734 cells
            on = [(a.clone(), b.clone()) for a, b in net(big)]
            net.set_latency_mode(False)
            off = net(big)
            assert all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(on, off))
    net.set_latency_mode(False)


def test_eval_val2017_harness_synthetic(dev, tmp_path):
    """tools/eval_val2017.py --synthetic: the val2017 evaluation (test.py:18-29 -> trainer/tester.py:26-52 -> eval/coco_eval.py:
    57-63,80-127) up to the two json files pycocotools would be handed -- keys, counts, one entry per detection in image order,
    boxes inside their images, and every RLE string decodable to exactly size[0] * size[1] pixels by the CPU restatement of
    pycocotools' string format (oracle/rle_ref.c).  No AP: neither COCO nor pycocotools exist offline (exit code 3 says so)."""
    import json
    import subprocess
    import sys
    out = tmp_path / "val"
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "eval_val2017.py"), "--synthetic", "4", "--batch", "4", "--out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    summary = json.load(open(out / "ap.json"))
    bbox = json.load(open(out / "bbox_prediction.json"))
    segm = json.load(open(out / "segm_prediction.json"))
    assert summary["images"] == 4 and summary["ranks"] == 1 and summary["detections"] == len(bbox) == len(segm) > 0
    assert summary["verdict"].startswith("synthetic dry run")
    assert [b["image_id"] for b in bbox] == sorted(b["image_id"] for b in bbox) and set(b["image_id"] for b in bbox) <= {0, 1, 2, 3}
    for b, s in zip(bbox, segm):
        assert set(b) == {"image_id", "category_id", "bbox", "score"} and set(s) == {"image_id", "category_id", "segmentation", "score"}
        assert b["image_id"] == s["image_id"] and b["category_id"] == s["category_id"] and b["score"] == s["score"]
        assert 1 <= b["category_id"] <= 90 and 0.0 < b["score"] <= 1.0 and len(b["bbox"]) == 4 and b["bbox"][2] >= 0 and b["bbox"][3] >= 0
        assert s["segmentation"]["size"] == [544, 544]
    for s in segm[:40]:
        runs = R.rle_string_decode(s["segmentation"]["counts"], 544 * 544)      # asserts the runs cover the image exactly
        assert all(c >= 0 for c in runs)


def test_bench_line_with_the_rccl_path_on_one_gpu(dev):
    """`bench.py` end to end as the driver runs it (a short one: 3 steps), with the N > 1 code path forced on ONE GPU
    (OM_BENCH_FORCE_DIST=1: nccl process group of one rank, weight broadcast, rank 0's solo reference run, per-rank reduction): the
    line carries BASELINE.json's metric, `roofline` with the like-for-like figures up front, and the multi-GPU fields of VERDICT
    round 4, item 8 -- at one rank `scaling_efficiency` is the timed region over the solo run of the same steps, i.e. ~1."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, OM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                        "--no-f32-compare", "--no-f16-compare", "--no-small-batch"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["metric"].startswith("images/sec end-to-end (544^2, bs=32)") and line["unit"] == "images/s" and line["n_gpus"] == 1
    assert line["rccl_ranks"] == 1 and line["value"] > 0 and line["one_batch_in_flight_value"] > 0
    assert len(line["per_rank_value"]) == 1 and abs(line["per_rank_value"][0] - line["value"]) < 1e-6 * line["value"] + 0.02
    assert line["solo_reference"]["value"] > 0 and 0.8 < line["scaling_efficiency"] < 1.25
    # ONE blob: the raw fp32 state_dict (~255 MB), packed per rank on the device
    assert line["weight_broadcast"]["blobs"] == 1 and 2.54e8 < line["weight_broadcast"]["bytes"] < 2.56e8 and line["weight_broadcast"]["gbs"] > 0
    assert line["weight_broadcast"]["blobs_identical_across_ranks"] is True
    rf = line["roofline"]
    assert list(rf)[:9] == ["bound", "achieved", "peak", "unit", "frac", "traffic", "frac_counts", "achieved_executed", "executed_frac"]
    # `frac` counts ALGORITHMIC (direct-convolution) flops; the fused F(4,3) kernel with split operands executes 1.5x as many
    assert rf["kernel"].startswith("wino14_split_kernel") and 0 < rf["frac"] < rf["executed_frac"] < 1
    assert abs(rf["executed_frac"] / rf["frac"] - 1.5) < 0.01 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert rf["one_batch_in_flight_images_per_s"] == line["one_batch_in_flight"]["value"]
