"""CPU: host logic and the C-ABI surface (no GPU compute calls)."""
import ctypes
import functools
import os
import re

import pytest
import torch

from conftest import REPO, post_cfg
from orienmask_amd import arch, lib as omlib, pack, synth


def test_library_exports_every_declared_symbol(built):
    """Every function include/orienmask_hip.h declares is exported and bound."""
    header = open(os.path.join(REPO, "include", "orienmask_hip.h")).read()
    declared = set(re.findall(r"\b(om_[a-z0-9_]+)\s*\(", header))
    declared -= {"om_model", "om_stream"}
    assert declared == set(omlib.SIGNATURES), declared ^ set(omlib.SIGNATURES)
    L = omlib.load()
    for name in declared:
        assert hasattr(L, name)
    assert L.om_version() == 140


def test_struct_layouts_match_header(built):
    assert ctypes.sizeof(omlib.LayerInfo) == 64 + 8 * 4 + 10 * 8
    assert ctypes.sizeof(omlib.PostCfg) == 4 * (1 + 3 + 3 + 2 + 1 + 9 + 9 + 9 + 1 + 2 + 2 + 1 + 1 + 2 + 3)
    assert ctypes.sizeof(omlib.RleImage) == 8 + 4 * 11 + 4          # pointer, eleven int32, tail padding to 8


def test_graph_matches_reference_state_dict(built):
    """90 convolutions, 524 keys, 63,662,063 parameters."""
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(num_anchors=3, num_classes=80, pretrained=None, freeze_backbone=False,
                               backbone_batchnorm_eval=False)
    sd = net.state_dict()
    assert len(sd) == 524
    assert sum(p.numel() for p in net.parameters()) == 63662063
    for k in ("backbone.conv1.conv_block.0.weight", "backbone.conv2.1.conv.0.conv_block.1.running_var",
              "bbox_head32.1.bias", "orien_head.5.weight", "skip4.conv_block.1.num_batches_tracked"):
        assert k in sd
    h = net._ensure_handle()
    assert len(net._layers) == 90 == len(arch.fpnplus_convs())
    L = omlib.load()
    assert L.om_forward_workspace_bytes(h, 1, 544, 544) > 0
    assert L.om_forward_workspace_bytes(h, 1, 540, 544) == 0          # not a multiple of 32
    # strict loading of a reference-format checkpoint, both accepted wrappers
    ref_sd = synth.synth_state_dict(3)
    net.load_state_dict(ref_sd, strict=True)
    with pytest.raises(RuntimeError):
        bad = dict(ref_sd); bad.pop("neck8.2.conv_block.0.weight")
        net.load_state_dict(bad, strict=True)
    assert pack.unwrap_checkpoint({"state_dict": ref_sd, "config": {}}) is ref_sd


def test_pack_folds_batchnorm(built):
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80)
    sd = synth.synth_state_dict(4)
    net.load_state_dict(sd)
    h = net._ensure_handle()
    L = omlib.load()
    blob = pack.pack_state_dict(net.state_dict(), net._layers, L.om_model_weight_floats(h))
    by_name = {l["name"]: l for l in net._layers}
    l = by_name["backbone.conv3.1.conv.1"]
    g, b = sd["backbone.conv3.1.conv.1.conv_block.1.weight"].double(), sd["backbone.conv3.1.conv.1.conv_block.1.bias"].double()
    m, v = sd["backbone.conv3.1.conv.1.conv_block.1.running_mean"].double(), sd["backbone.conv3.1.conv.1.conv_block.1.running_var"].double()
    scale = g / torch.sqrt(v + 1e-5)
    assert torch.allclose(blob[l["scale_off"]:l["scale_off"] + l["cout"]].double(), scale, rtol=1e-6)
    assert torch.allclose(blob[l["shift_off"]:l["shift_off"] + l["cout"]].double(), b - m * scale, rtol=1e-5, atol=1e-7)
    w = sd["backbone.conv3.1.conv.1.conv_block.0.weight"]
    got = blob[l["w_off"]:l["w_off"] + w.numel()].view(128, 3, 3, 64)
    assert torch.equal(got, w.permute(0, 2, 3, 1))
    # bias-only head: scale 1, shift = bias, rows >= cout zero
    l = by_name["bbox_head16.1"]
    assert l["cout"] == 255 and l["cout_pad"] == 256 and not l["has_bn"]
    assert torch.equal(blob[l["scale_off"]:l["scale_off"] + 256], torch.cat([torch.ones(255), torch.zeros(1)]))
    assert torch.equal(blob[l["shift_off"]:l["shift_off"] + 255], sd["bbox_head16.1.bias"])
    assert blob[l["w_off"] + 255 * 512:l["w_off"] + 256 * 512].abs().sum() == 0
    assert l["wino_off"] == -1
    # Winograd weights of a stride-1 3x3 layer: U = G_y g G_x^T.  16 planes (F(2x2,3x3)): xi = 0 is g[0][0], xi = 15 is
    # g[2][2], xi = 5 is the sum of all nine taps / 4.  24 planes (F(2x4,3x3)): xi = 0 is g[0][0] / 4, xi = 23 is g[2][2],
    # xi = 6 + 1 is -(sum of all nine taps) / 12.
    l = by_name["backbone.conv3.1.conv.1"]
    assert l["wino_off"] >= 0 and by_name["backbone.conv3.0"]["wino_off"] == -1 and by_name["backbone.conv1"]["wino_off"] == -1
    assert by_name["backbone.conv3.0"]["wino_planes"] == 0 and l["wino_planes"] in (16, 24)
    planes = l["wino_planes"]
    u = blob[l["wino_off"]:l["wino_off"] + planes * 128 * 64].view(planes, 128, 64)
    if planes == 16:
        assert torch.equal(u[0], w[:, :, 0, 0]) and torch.equal(u[15], w[:, :, 2, 2])
        assert torch.allclose(u[5], w.double().sum((2, 3)).float() / 4, rtol=1e-6, atol=1e-8)
    else:
        assert torch.equal(u[0], w[:, :, 0, 0] / 4) and torch.equal(u[23], w[:, :, 2, 2])
        assert torch.allclose(u[7], -(w.double().sum((2, 3)) / 12).float(), rtol=1e-6, atol=1e-8)
    # every stride-1 3x3 layer carries both transforms: F(2x4) for full batches, F(2x2) (wino_alt_off) for a few images
    l6 = by_name["backbone.conv6.1.conv.1"]
    assert l6["wino_planes"] == 24 and l6["wino_alt_off"] > l6["wino_off"]
    u22 = blob[l6["wino_alt_off"]:l6["wino_alt_off"] + 16 * 1024 * 512].view(16, 1024, 512)
    w6 = sd["backbone.conv6.1.conv.1.conv_block.0.weight"]
    assert torch.equal(u22[0], w6[:, :, 0, 0]) and torch.equal(u22[15], w6[:, :, 2, 2])


def test_registry_builders_mirror_reference(built):
    """build()/build_postprocess() consume the reference's config dicts unchanged
    (/root/reference/trainer/builder.py:61-77, /root/reference/config/base.py:219-236)."""
    from orienmask_amd import builder, eval as om_eval, model as om_model
    cfg = dict(type="OrienMaskYOLOPostProcess", nms=dict(type="batched_nms", threshold=0.45), **post_cfg((544, 544)))
    keep = dict(cfg)
    post = builder.build_postprocess(cfg, device=torch.device("cpu"))
    assert cfg == keep                                      # the caller's dict is not mutated
    assert isinstance(post, om_eval.OrienMaskYOLOPostProcess)
    assert post.nms_thresh == pytest.approx(0.45) and post.nms_pre == 400 and post.nms_post == 100
    assert isinstance(post.nms, functools.partial) and post.nms.func is om_eval.batched_nms
    net = builder.build(dict(type="OrienMaskYOLOFPNPlus", num_anchors=3, num_classes=80, pretrained=None,
                             freeze_backbone=False, backbone_batchnorm_eval=False), om_model)
    assert isinstance(net, om_model.OrienMaskYOLOFPNPlus)
    # any callable is accepted as nms_func, like the reference's (postprocess.py:9-11): it runs between the device-side
    # candidate and mask stages instead of inside the fused kernel (tests/test_hip_parity.py covers the results)
    foreign = om_eval.OrienMaskYOLOPostProcess(nms_func=lambda d, c: None, **post_cfg((544, 544)))
    assert foreign.nms_thresh is None and foreign.cfg_struct(256).nms_thresh == pytest.approx(0.5)
    # 1..3 scales, 1..3 anchors each; the library's tables end there and say so
    two = om_eval.OrienMaskYOLOPostProcess([[3, 4], [6, 8]], [96, 128], post_cfg((96, 128))["anchors"], [[6, 7, 8], [3, 4]], 80)
    c2 = two.cfg_struct(256)
    assert (c2.num_scales, list(c2.anchors_of_scale)[:2], list(c2.anchor_mask[1])[:2]) == (2, [3, 2], [3, 4])
    for bad_masks, bad_grids in (([[0, 1, 2, 3]], [[3, 4]]), ([[0], [1], [2], [3]], [[3, 4]] * 4), ([[0, 11]], [[3, 4]])):
        with pytest.raises(ValueError):
            om_eval.OrienMaskYOLOPostProcess(bad_grids, [96, 128], post_cfg((96, 128))["anchors"], bad_masks, 80)
    c = post.cfg_struct(256)
    assert (c.grid_h[0], c.grid_w[2], c.image_h, c.anchors_per_scale, c.num_classes) == (17, 68, 544, 3, 80)
    assert list(c.anchor_mask[0]) == [6, 7, 8] and c.anchor_w[8] == 459.0 and c.anchor_h[8] == 401.0
    L = omlib.load()
    assert L.om_postprocess_workspace_bytes(ctypes.byref(c), 32) > 32 * 18207 * 80 * 4
    c.nms_pre = 4096
    assert L.om_postprocess_workspace_bytes(ctypes.byref(c), 1) == 0     # beyond the fused path's limit


def test_builder_optional_nms_and_backends(built):
    """trainer/builder.py:75 pops 'nms' with a default of None (-> the default batched_nms); the NMS backend and
    batched_nms(normalized=...) travel from the config into the kernel's cfg struct."""
    from orienmask_amd import builder, eval as om_eval
    post = builder.build_postprocess(dict(type="OrienMaskYOLOPostProcess", **post_cfg((544, 544))), device=torch.device("cpu"))
    assert post.nms is om_eval.batched_nms and post.nms_thresh == 0.5 and post.nms_backend == "cpu" and post.nms_normalized
    c = post.cfg_struct(256)
    assert (c.nms_semantics, c.nms_normalized) == (0, 1)
    post = builder.build_postprocess(dict(type="OrienMaskYOLOPostProcess",
                                          nms=dict(type="batched_nms", threshold=0.6, normalized=False, backend="cuda"),
                                          **post_cfg((544, 544))), device=torch.device("cpu"))
    c = post.cfg_struct(256)
    assert (c.nms_semantics, c.nms_normalized, post.nms_thresh) == (1, 0, pytest.approx(0.6))
    om_eval.set_nms_backend("cuda")
    try:
        assert om_eval.OrienMaskYOLOPostProcess(**post_cfg((96, 96))).nms_backend == "cuda"
        assert om_eval.OrienMaskYOLOPostProcess(nms_backend="cpu", **post_cfg((96, 96))).nms_backend == "cpu"
    finally:
        om_eval.set_nms_backend("cpu")
    with pytest.raises(ValueError):
        om_eval.set_nms_backend("rocm")
    with pytest.raises(ValueError):
        om_eval.OrienMaskYOLOPostProcess(**dict(post_cfg((96, 96)), nms_pre=2000))
    with pytest.raises(TypeError):
        om_eval.OrienMaskYOLOPostProcess(nms_func=functools.partial(om_eval.batched_nms, iou=0.5), **post_cfg((96, 96)))


def test_pretrained_loads_in_the_backbone_key_space(tmp_path, built):
    """model/base.py:48-64: the pretrained file is loaded INSIDE the backbone, so its keys are backbone-relative
    ('conv1.conv_block.0.weight', as in pretrained_darknet53.pth); foreign / mis-shaped keys are ignored and reported."""
    import warnings
    from orienmask_amd import model as om_model
    sd = synth.synth_state_dict(13)
    backbone = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    backbone["fc.weight"] = torch.zeros(1000, 1024)                        # classifier head of the ImageNet checkpoint
    backbone["conv1.conv_block.0.weight"] = torch.zeros(32, 3, 5, 5)       # wrong shape: ignored
    path = str(tmp_path / "pretrained_darknet53.pth")
    torch.save(backbone, path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net = om_model.OrienMaskYOLOFPNPlus(3, 80, pretrained=path)
    assert any("ignored keys" in str(x.message) for x in w)
    own = net.state_dict()
    assert torch.equal(own["backbone.conv6.4.conv.1.conv_block.0.weight"], sd["backbone.conv6.4.conv.1.conv_block.0.weight"])
    assert torch.equal(own["backbone.conv2.0.conv_block.1.running_var"], sd["backbone.conv2.0.conv_block.1.running_var"])
    assert not torch.equal(own["backbone.conv1.conv_block.0.weight"], sd["backbone.conv1.conv_block.0.weight"])
    assert not torch.equal(own["neck32.0.conv_block.0.weight"], sd["neck32.0.conv_block.0.weight"])
    torch.save({"backbone." + k: v for k, v in backbone.items()}, path)    # full-model keys do NOT match (as in the reference)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        om_model.OrienMaskYOLOFPNPlus(3, 80, pretrained=path)
    assert any("none of its" in str(x.message) for x in w)


def test_cpu_tensors_are_rejected_loudly(built):
    from orienmask_amd import eval as om_eval
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80).eval()
    with pytest.raises(omlib.OrienMaskHipError):
        net(torch.zeros(1, 3, 64, 64))
    with pytest.raises(omlib.OrienMaskHipError):
        om_eval.batched_nms(torch.rand(4, 5), torch.zeros(4, dtype=torch.long))
    post = om_eval.OrienMaskYOLOPostProcess(**post_cfg((96, 96)))
    heads = synth.synth_heads(1, 1, post_cfg((96, 96))["grid_size"])
    with pytest.raises(omlib.OrienMaskHipError):
        post(heads)


def test_in_flight_pipeline_host_logic(built):
    """The pipeline's bookkeeping without a GPU: argument checks, slot-local postprocess workspaces, workspace_slot scoping."""
    from orienmask_amd import eval as om_eval
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    from orienmask_amd.pipeline import InFlightPipeline
    net = OrienMaskYOLOFPNPlus(3, 80).eval()
    post = om_eval.OrienMaskYOLOPostProcess(**post_cfg((96, 96)))
    with pytest.raises(ValueError):
        InFlightPipeline(net, post, depth=0)
    pipe = InFlightPipeline(net, post, depth=3)
    # every batch in flight has a PRIVATE postprocess copy (the caller's instance stays the eager path's)
    assert all(p is not post for p in pipe._posts) and len({id(p._ws) for p in pipe._posts} | {id(post._ws)}) == 4
    assert all(p.cfg_struct(256).nms_pre == post.nms_pre for p in pipe._posts)
    with pytest.raises(omlib.OrienMaskHipError):
        pipe.submit(torch.zeros(1, 3, 96, 96))            # CPU tensor: no fallback
    assert len(pipe) == 0
    with pytest.raises(RuntimeError):
        pipe.result()
    assert net._slot == 0
    with net.workspace_slot(2):
        assert net._slot == 2
        with pytest.raises(ValueError):
            with net.workspace_slot(-1):
                pass
        assert net._slot == 2
    assert net._slot == 0


def test_missing_library_is_fatal(monkeypatch, built):
    monkeypatch.setattr(omlib, "_lib", None)
    monkeypatch.setattr(omlib, "LIB_PATH", "/nonexistent/liborienmask_hip.so")
    with pytest.raises(omlib.OrienMaskHipError):
        omlib.load()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "orienmask_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "nms_ref" not in text and "libnms_ref" not in text and "nms_cuda_ref" not in text, f


def test_timer_api_mirrors_reference():
    """utils/timer.py semantics: named timers, mean over uses, cpu()/cuda() switch, reset()."""
    import time
    from orienmask_amd import timer
    timer.reset(); timer.cpu()
    for _ in range(3):
        with timer.timer("stage a"):
            time.sleep(0.002)
    with timer.timer("stage b"):
        pass
    log = timer.get_all_elapsed_time()
    assert list(log) == ["stage a", "stage b"] and log["stage a"] >= 1.5 and log["stage b"] < 1.0
    timer.reset()
    assert timer.get_all_elapsed_time() == {}
    timer.cuda()


def test_transform_registry_and_pad_info(built):
    from orienmask_amd import transform as T
    cfg = dict(type="FastCOCOTransform", pipeline=[dict(type="Resize", size=(544, 544), interpolation="bilinear", align_corners=False),
                                                   dict(type="Normalize", mean=(0, 0, 0), std=(255, 255, 255))])
    keep = {k: (list(v) if isinstance(v, list) else v) for k, v in cfg.items()}
    tf = T.build_transform(cfg)                       # /root/reference/config/base.py:158-164 goes in unchanged
    assert cfg["pipeline"] == keep["pipeline"] and isinstance(tf._resize, T.FastCOCOTransform.Resize)
    assert tf._resize.size == (544, 544) and tf._norm.std == [255.0] * 3
    assert T.FastCOCOTransform.ShortEdgeResize(544, 800).target(480, 640) == (544, 725)
    with pytest.raises(NotImplementedError):
        T.FastCOCOTransform.Resize((544, 544), interpolation="nearest")
    with pytest.raises(omlib.OrienMaskHipError):
        tf(torch.zeros(1, 8, 8, 3))                   # CPU tensor: no fallback


def test_checkpoint_ingest_reference_format(tmp_path, built):
    """a .pth as the reference's trainer writes it (trainer/base.py:143-152) -> packed blob."""
    from orienmask_amd import builder, model as om_model
    sd = synth.synth_state_dict(12)
    cfg = dict(model=dict(type="OrienMaskYOLOFPNPlus", num_anchors=3, num_classes=80, pretrained="checkpoints/x.pth",
                          freeze_backbone=False, backbone_batchnorm_eval=False))
    ckpt = dict(epoch=100, state_dict=sd, optimizer={}, lr_scheduler={}, monitor_best=0.345, config=cfg)
    path = str(tmp_path / "best_model.pth")
    torch.save(ckpt, path)
    got_sd, got_cfg = builder.load_checkpoint(path)
    assert got_cfg == cfg and set(got_sd) == set(sd)
    raw = str(tmp_path / "weights_only.pth")
    torch.save(sd, raw)                                     # infer.py also accepts a bare state_dict
    got_sd2, none_cfg = builder.load_checkpoint(raw)
    assert none_cfg is None and all(torch.equal(got_sd2[k], sd[k]) for k in sd)
    net = om_model.OrienMaskYOLOFPNPlus(3, 80)
    net.load_state_dict(got_sd, strict=True)
    h = net._ensure_handle()
    total = omlib.load().om_model_weight_floats(h)
    a = pack.pack_state_dict(net.state_dict(), net._layers, total)
    b = pack.pack_state_dict({"state_dict": sd}, net._layers, total)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        builder.build_tester(dict(postprocess={}), raw, [], device=torch.device("cpu"))


def test_split_f16_pairs_layout_and_accuracy():
    """pack.split_f16_pairs / winograd_weights_split: the [16 hi | 16 lo] row layout of include/orienmask_hip.h
    (om_layer_info.wsplit_off), hi + lo reproduces the scaled fp32 weights to 2^-21 of the output channel's largest element,
    the scaling is a power of two per output channel and keeps fp16 far from overflow; and the three-product form
    hi*hi + hi*lo + lo*hi of a dot product is closer to the float64 answer than fp32 accumulation error."""
    import torch
    from orienmask_amd.pack import split_f16_pairs, winograd_weights, winograd_weights_split
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 32, generator=g) * 7
    p = split_f16_pairs(x)
    assert p.shape == (3, 2, 2, 16) and p.dtype == torch.float16
    assert torch.equal(p[:, :, 0].reshape(3, 32), x.half())
    assert torch.equal(p[:, :, 1].reshape(3, 32), (x - x.half().float()).half())
    w = torch.randn(70, 64, 3, 3, generator=g) / 24
    w[5] *= 1e-3; w[6] *= 300.0; w[7] = 0
    us, e = winograd_weights_split(w, 128)
    assert us.shape == (24, 128, 4, 2, 16) and e.shape == (128,) and int(e[7]) == 0 and int(e[100]) == 0
    u = winograd_weights(w, 128, 24).double()
    back = (us[..., 0, :].double() + us[..., 1, :].double()).reshape(24, 128, 64) * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e.double()).view(1, -1, 1)
    amax = u.abs().amax(dim=(0, 2)).clamp_min(1e-30)
    assert float(((back - u).abs().amax(dim=(0, 2)) / amax).max()) < 2.0 ** -21
    assert float(us.float().abs().max()) < 2.0 ** 14
    v = torch.randn(512, 64, generator=g) * 3
    vp = split_f16_pairs(v).double()
    vh, vl = vp[:, :, 0].reshape(512, 64), vp[:, :, 1].reshape(512, 64)
    uh, ul = us[3, :70, :, 0].reshape(70, 64).double(), us[3, :70, :, 1].reshape(70, 64).double()
    want = v.double() @ u[3, :70].T
    got = (vh @ uh.T + vh @ ul.T + vl @ uh.T) * torch.pow(torch.tensor(2.0, dtype=torch.float64), -e[:70].double()).view(1, -1)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) / scale < 2e-7
    assert float(((v @ u[3, :70].float().T).double() - want).abs().max()) / scale > float((got - want).abs().max()) / scale


def test_pack_state_dict_split_covers_every_layer(built):
    """The split blob (om_model_load_weights_split): every layer but the stem has a slice, slices do not overlap and fill the
    blob up to alignment, and hi + lo times the per-channel scale reproduces the fp32 blob's weights (one 1x1 layer in the
    implicit-GEMM order, one stride-1 3x3 layer in the F(2x4) order) and folded BatchNorm scales."""
    import torch
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    net = OrienMaskYOLOFPNPlus(3, 80)
    sd = synth.synth_state_dict(4)
    net.load_state_dict(sd, strict=True)
    h = net._ensure_handle()
    L = omlib.load()
    layers = net._layers
    total = L.om_model_weight_split_words(h)
    blob32 = pack.pack_state_dict(sd, layers, L.om_model_weight_floats(h))
    blob = pack.pack_state_dict_split(sd, layers, total)
    assert blob.numel() == total and torch.isfinite(blob).all()
    spans = []
    for l in layers:
        if l["name"] == "backbone.conv1":
            assert l["wsplit_off"] == -1
            continue
        n = (18 if l["wino_planes"] == 24 else l["ksize"] ** 2) * l["cout_pad"] * l["cin"]
        spans.append((l["wsplit_off"], l["wsplit_off"] + n))
        spans.append((l["wsplit_scale_off"], l["wsplit_scale_off"] + l["cout_pad"]))
        if l["wino_planes"] == 24:      # the latency mode's direct form of the same layer (om_model_set_latency_cells)
            spans.append((l["wsplit_direct_off"], l["wsplit_direct_off"] + 9 * l["cout_pad"] * l["cin"]))
            spans.append((l["wsplit_direct_scale_off"], l["wsplit_direct_scale_off"] + l["cout_pad"]))
            d = blob[l["wsplit_direct_off"]:l["wsplit_direct_off"] + 9 * l["cout_pad"] * l["cin"]].view(torch.float16)
            assert torch.isfinite(d.float()).all() and d.abs().max() >= 2.0 ** 13
        else:
            assert l["wsplit_direct_off"] == -1
    spans.sort()
    assert spans[0][0] == 0 and all(a[1] <= b[0] < a[1] + 4 for a, b in zip(spans, spans[1:])) and total - spans[-1][1] < 4
    perm = torch.tensor(pack._SPLIT_PERM)
    for name in ("backbone.conv4.1.conv.0", "neck8.1"):
        l = next(x for x in layers if x["name"] == name)
        cpad, cin, cout = l["cout_pad"], l["cin"], l["cout"]
        scale32 = blob32[l["scale_off"]:l["scale_off"] + cpad].double()
        scale_s = blob[l["wsplit_scale_off"]:l["wsplit_scale_off"] + cpad].double()
        ratio = (scale32[:cout] / scale_s[:cout])                      # = 2^e, exactly
        assert torch.equal(ratio, torch.pow(torch.tensor(2.0, dtype=torch.float64), torch.round(torch.log2(ratio))))
        if l["wino_planes"] == 24:
            # the fused F(4,3) form: [cout_pad/64][cin/16][6 j][3 ky][64][hi 8 | hi 8 | lo 8 | lo 8] (conv_wino14.hip)
            halfs = blob[l["wsplit_off"]:l["wsplit_off"] + 18 * cpad * cin].view(torch.float16).reshape(cpad // 64, cin // 16, 6, 3, 64, 2, 16).double()
            back = (halfs[..., 0, :] + halfs[..., 1, :]).permute(3, 2, 0, 4, 1, 5).reshape(3, 6, cpad, cin)[:, :, :cout] / ratio.view(1, 1, -1, 1)
            w4 = sd[name + ".conv_block.0.weight"].double()
            want = torch.einsum("js,ncrs->rjnc", pack._WINO_G6, w4)
        else:
            k2 = l["ksize"] ** 2
            halfs = blob[l["wsplit_off"]:l["wsplit_off"] + k2 * cpad * cin].view(torch.float16).reshape(cpad, k2, cin // 16, 2, 2, 8).double()
            grp = torch.empty(cpad, k2, cin // 16, 16, dtype=torch.float64)
            grp[..., perm] = (halfs[..., 0, :, :] + halfs[..., 1, :, :]).reshape(cpad, k2, cin // 16, 16)
            back = grp.reshape(cpad, k2 * cin)[:cout] / ratio.view(-1, 1)
            want = blob32[l["w_off"]:l["w_off"] + cout * k2 * cin].reshape(cout, k2 * cin).double()
        amax = want.abs().amax()
        assert float((back - want).abs().max() / amax) < 2.0 ** -20, name


def test_upsample_on_read_graph_logic(built):
    """Host side of the up-sampling-on-read form (om_model.cpp: find_gathers), no GPU: in split-operand mode the three 1x1 layers
    behind the reference's up-sample + concat (orienmask_yolo_fpnplus.py:78-86) report the gather kernel and nothing else does;
    the switch, the other precision and kept activations select the replicated form; the workspace size is answered in both
    forms (the side buffers are small enough to fall into gaps of the live-range layout at this size); the non-Plus model has
    the same three (its neck4.0 sits behind route8 + x4)."""
    from orienmask_amd.model import OrienMaskYOLO, OrienMaskYOLOFPNPlus
    L = omlib.load()
    net = OrienMaskYOLOFPNPlus(3, 80).set_precision("f32_split")
    h = net._ensure_handle()
    kinds = dict(net.layer_kernels(32, 544, 544))
    gathered = sorted(k for k, v in kinds.items() if "gather" in v)
    assert gathered == ["neck16.0", "neck4.0", "neck8.0"]
    ws_gather = L.om_forward_workspace_bytes(h, 32, 544, 544)
    net.set_upsample_on_read(False)
    assert not any("gather" in v for _, v in net.layer_kernels(32, 544, 544))
    ws_replicated = L.om_forward_workspace_bytes(h, 32, 544, 544)
    assert ws_gather > 0 and ws_replicated > 0
    net.set_upsample_on_read(True)
    assert L.om_forward_workspace_bytes(h, 32, 544, 544) == ws_gather
    net.keep_activations(True)
    assert not any("gather" in v for _, v in net.layer_kernels(32, 544, 544))
    net.keep_activations(False)
    net.set_precision("f32")
    assert not any("gather" in v for _, v in net.layer_kernels(32, 544, 544))
    small = OrienMaskYOLO(3, 80).set_precision("f32_split")
    assert sorted(k for k, v in small.layer_kernels(8, 544, 544) if "gather" in v) == ["neck16.0", "neck4.0", "neck8.0"]
    assert L.om_model_set_upsample_on_read(None, 1) != 0          # null model: an error code, not a crash


def test_wino14d_isa_audit(tmp_path):
    """conv_wino14d.hip owns the accumulation registers a0 .. a191 by name (inline-asm matrix instructions); the compiler knows them
    only as clobbers.  That is sound only while the compiler has no reason to touch them and the asm needs no padding it does
    not get -- checked on the emitted gfx950 code (hipcc cross-compiles here, no GPU): no scratch, no spilled register, no
    compiler-issued v_accvgpr_* at all, exactly 192 accumulation registers, and no vector-ALU write of a matrix instruction's
    operand within the two instructions in front of it (the wait states the compiler does not insert inside an asm)."""
    src = os.path.join(REPO, "orienmask_amd", "csrc", "conv_wino14d.hip")
    out = str(tmp_path / "w14d.s")
    import subprocess
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src, "-o", out,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    remarks = r.stderr
    kernels = re.findall(r"Function Name: (\S*wino14_dual_kernel\S*)", remarks)
    assert len(kernels) == 2
    assert re.findall(r"VGPRs Spill: (\d+)", remarks) == ["0", "0"], remarks
    assert re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", remarks) == ["0", "0"], remarks
    assert re.findall(r"AGPRs: (\d+)", remarks) == ["192", "192"], remarks
    text = open(out).read()
    for name in kernels:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        in_asm = False
        code = []           # (instruction, inside an asm statement)
        for raw in body.split("\n"):
            if "#ASMSTART" in raw:
                in_asm = True
                continue
            if "#ASMEND" in raw:
                in_asm = False
                continue
            ins = raw.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            code.append((ins, in_asm))
        assert sum(1 for ins, a in code if ins.startswith("v_mfma")) >= 216
        for i, (ins, a) in enumerate(code):
            if "accvgpr" in ins:
                assert a, "compiler-issued %s" % ins
            if ins.startswith("v_mfma"):
                assert a
                ops = set()
                for m in re.finditer(r"v\[(\d+):(\d+)\]", ins):
                    ops.update(range(int(m.group(1)), int(m.group(2)) + 1))
                for prev, _ in code[max(0, i - 2):i]:
                    if prev.startswith("v_") and not prev.startswith("v_mfma") and not prev.startswith("v_cmp"):
                        d = re.match(r"\S+\s+v\[?(\d+)(?::(\d+))?\]?", prev)
                        if d:
                            lo, hi = int(d.group(1)), int(d.group(2) or d.group(1))
                            assert not (ops & set(range(lo, hi + 1))), (prev, ins)
    del lines


def test_no_packed_fp32_register_half_select(tmp_path):
    """gfx950 erratum found in round 5 (tools/hazard_probe/pk_opsel_repro.hip, profiles/r05_experiments.md 2): v_pk_add/mul/fma_f32
    with a source-half selection (op_sel / op_sel_hi) on a REGISTER operand returns wrong lanes now and then while another wave on
    the same SIMD issues wide-K matrix instructions -- which this library's kernels do on every other stream.  hipcc emits the form
    when it packs scalar code (SLP) or broadcasts a scalar into vector arithmetic.  No kernel of the library may contain one.  Since
    round 6 the BUILD enforces it (csrc/Makefile: every file's gfx950 code is kept by -save-temps=obj and scanned by
    csrc/isa_audit.py, which fails the rule; -fno-slp-vectorize on the files whose code had the form); this test checks that the rule is there for every
    source file, that the scanner flags the form and nothing else, and re-runs it over the code of the library that was built."""
    import subprocess
    import importlib.util
    csrc = os.path.join(REPO, "orienmask_amd", "csrc")
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(csrc, "isa_audit.py"))
    audit = importlib.util.module_from_spec(spec); spec.loader.exec_module(audit)
    sample = tmp_path / "sample.s"
    sample.write_text("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[0,1,1]\n"          # a register's low half twice: the erratum's form
                      "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]\n"                        # the high half of a register
                      "\tv_pk_mul_f32 v[0:1], v[2:3], 1.0 op_sel_hi:[1,0]\n"                        # a half-selected constant: fine
                      "\tv_pk_add_f32 v[0:1], v[2:3], s[4:5] op_sel_hi:[1,0]\n"                     # a scalar pair: fine
                      "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]\n")                            # no selection: fine
    assert len(audit.bad_instructions(str(sample))) == 2
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert re.search(r"^CXXFLAGS = .*-save-temps=obj", mk, re.M) and re.search(r"^NOPK = -fno-slp-vectorize$", mk, re.M)
    assert re.search(r"^build/%\.o:.*isa_audit\.py", mk, re.M) and "\t$(PYTHON) isa_audit.py build/$*-hip-amdgcn-amd-amdhsa-$(ARCH).s" in mk
    r = subprocess.run(["make", "-C", csrc, "audit"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    files = sorted(f[:-4] for f in os.listdir(csrc) if f.endswith(".hip") and f != "conv_wino14d.hip")
    assert len(files) >= 12
    for f in files:     # every product source file's code was there to be scanned
        assert os.path.exists(os.path.join(csrc, "build", f + "-hip-amdgcn-amd-amdhsa-gfx950.s")), f
    # the dual-role kernel is not in the default library; its code is held to the same rule
    out = str(tmp_path / "w14d.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                        "--cuda-device-only", os.path.join(csrc, "conv_wino14d.hip"), "-o", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert audit.bad_instructions(out) == []


def test_bench_clock_sampler_reads_the_drivers_table(tmp_path):
    """bench.ClockSampler: the level marked `*` of a pp_dpm_sclk table is the current shader clock; without a table (no GPU, no sysfs
    access) it samples nothing and reports None -- the bench line then simply has no `sclk_mhz`."""
    import bench
    s = bench.ClockSampler.__new__(bench.ClockSampler)
    import threading
    s._threading, s.samples, s._stop, s._thread = threading, [], None, None
    table = tmp_path / "pp_dpm_sclk"
    table.write_text("0: 500Mhz\n1: 1987Mhz *\n2: 2400Mhz\n")
    s.path = str(table)
    assert s._read() == 1987.0
    out = s.start().stop()
    assert out is not None and out["mean"] == 1987.0 and out["samples"] >= 1
    s.path = None
    assert s.start().stop() is None


def test_blob_checksum_tells_blobs_apart():
    """orienmask_amd.dist.blob_checksum (round 6: every rank packs the broadcast weights itself and the ranks all-gather a checksum of
    each packed blob): equal bytes -> equal checksums; one changed word, two swapped words or a changed fp16 half -> another."""
    import torch
    from orienmask_amd.dist import blob_checksum
    g = torch.Generator().manual_seed(7)
    a = torch.randn(100003, generator=g)
    assert blob_checksum(a) == blob_checksum(a.clone())
    b = a.clone(); b[5] = b[5] + 1e-3
    c = a.clone(); c[[10, 11]] = c[[11, 10]]
    assert blob_checksum(b) != blob_checksum(a) and blob_checksum(c) != blob_checksum(a)
    h = torch.randn(4099, generator=g).half()
    h2 = h.clone(); h2[77] = h2[77] + 1
    assert blob_checksum(h) == blob_checksum(h.clone()) and blob_checksum(h2) != blob_checksum(h)
