"""Generate tests/golden/*.npz by running the REAL reference from /root/reference.

Runs ONLY in the build container (the reference does not exist on the GPU box and never
travels).  The fixtures are data: seeds, inputs and the reference's outputs.  Weights and
head tensors are regenerated from seeds by orienmask_amd.synth, so the files stay small.

Recipe: never write bytecode into /root/reference; stub the
reference's unavailable imports (torchsummary, pycocotools); inject the reference's own
nms_cpu.cpp, compiled by oracle/build_ref.py, as eval.nms_cpu; eval.nms_cuda is an empty
stub that CPU tensors never reach.

    python tools/gen_golden.py            # writes tests/golden/
"""
import functools
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from orienmask_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
ANCHORS_YOLOV4 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]


def import_reference():
    for name in ("torchsummary", "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools.coco"].COCO = object
    sys.modules["pycocotools.cocoeval"].COCOeval = object
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    import nms_cpu_ref
    sys.modules["eval.nms_cpu"] = nms_cpu_ref
    sys.modules["eval.nms_cuda"] = types.ModuleType("eval.nms_cuda")
    sys.path.insert(0, "/root/reference")
    import config as ref_config
    import model as ref_model
    import eval as ref_eval
    import eval.function as ref_function
    ref_eval.nms_cpu = nms_cpu_ref
    return ref_config, ref_model, ref_eval, ref_function


class single_thread:
    """The reference's postprocess under ONE torch thread.  torch's CPU sigmoid is not bit-reproducible across thread
    counts: TensorIterator hands each thread a LINEAR element range, a range that starts or ends inside a row of class logits
    is processed as a shorter row, and which of a row's elements fall in the vectorised part (Sleef expf) and which in the
    scalar tail (glibc expf) then shifts -- the two differ by one ulp on ~4 % of inputs.  With one thread every row is whole,
    which is also what csrc/ref_math.h restates; so the fixtures are the reference's answer as `OMP_NUM_THREADS=1` gives it."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


def pack_masks(m):
    m = np.asarray(m, dtype=np.uint8)
    return np.packbits(m.reshape(m.shape[0], int(np.prod(m.shape[1:]))), axis=1)


def digest(t, nsamp=64):
    a = t.detach().cpu().double().reshape(-1)
    idx = torch.linspace(0, a.numel() - 1, nsamp).long()
    return np.array([a.sum().item(), a.abs().sum().item()]), a[idx].float().numpy(), idx.numpy()


def post_cfg(size_hw):
    h, w = size_hw
    return dict(grid_size=[[h // 32, w // 32], [h // 16, w // 16], [h // 8, w // 8]], image_size=[h, w],
                anchors=ANCHORS_YOLOV4, anchor_mask=ANCHOR_MASK, num_classes=80, conf_thresh=0.005,
                nms_pre=400, nms_post=100, orien_thresh=0.3)


def main():
    os.makedirs(OUT, exist_ok=True)
    cfg, rmodel, reval, rfunc = import_reference()
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- G4: NMS known answers
    nms_cases = {}
    rng = np.random.Generator(np.random.PCG64(77))

    def run_nms(name, dets, cats, thr=0.5):
        d = torch.from_numpy(np.asarray(dets, dtype=np.float32).reshape(-1, 5))
        c = torch.from_numpy(np.asarray(cats, dtype=np.int64).reshape(-1))
        kd, kc, keep = rfunc.batched_nms(d, c, threshold=thr)
        nms_cases[name + "_dets"] = d.numpy(); nms_cases[name + "_cats"] = c.numpy()
        nms_cases[name + "_thr"] = np.float32(thr); nms_cases[name + "_keep"] = keep.numpy()
        # plain (un-batched) nms on the same boxes through the native entry point
        nms_cases[name + "_keep_plain"] = sys.modules["eval.nms_cpu"].nms(d, thr).numpy() if d.shape[0] else np.zeros(0, np.int64)

    run_nms("empty", np.zeros((0, 5)), np.zeros(0))
    run_nms("single", [[0.5, 0.5, 0.2, 0.2, 0.9]], [3])
    # IoU exactly 0.5: boxes (0,0,2,1)-(1,0,2,1) shifted by 2/3 ... use w=3: overlap 2 of union 4
    run_nms("iou_half", [[1.5, 0.5, 3.0, 1.0, 0.9], [2.5, 0.5, 3.0, 1.0, 0.8], [8.0, 0.5, 3.0, 1.0, 0.7]], [0, 0, 0])
    run_nms("iou_half_strict", [[1.5, 0.5, 3.0, 1.0, 0.9], [2.5, 0.5, 3.0, 1.0, 0.8]], [0, 0], thr=0.5000001)
    run_nms("cross_class", [[0.5, 0.5, 0.4, 0.4, 0.9], [0.5, 0.5, 0.4, 0.4, 0.8], [0.52, 0.5, 0.4, 0.4, 0.7]], [1, 2, 1])
    run_nms("score_ties", [[0.5, 0.5, 0.4, 0.4, 0.5], [0.51, 0.5, 0.4, 0.4, 0.5], [0.9, 0.9, 0.1, 0.1, 0.5],
                           [0.52, 0.5, 0.4, 0.4, 0.5]], [0, 0, 0, 0])
    for n in (63, 64, 65, 200, 400):
        ctr = rng.random((n, 2)) * 0.8 + 0.1
        wh = rng.random((n, 2)) * 0.25 + 0.03
        sc = rng.random((n, 1))
        run_nms("rand%d" % n, np.concatenate([ctr, wh, sc], 1), rng.integers(0, 4, n))
    # clustered: many near-duplicates
    base = rng.random((20, 4)) * np.array([0.6, 0.6, 0.2, 0.2]) + np.array([0.2, 0.2, 0.1, 0.1])
    dup = np.repeat(base, 15, 0) + rng.standard_normal((300, 4)) * 0.01
    run_nms("clustered300", np.concatenate([dup, rng.random((300, 1))], 1), rng.integers(0, 2, 300))
    np.savez_compressed(os.path.join(OUT, "nms_kat.npz"), **nms_cases)
    print("nms_kat: %d cases" % (len([k for k in nms_cases if k.endswith("_keep")])))

    # ---------------------------------------------------------------- G3: postprocess on synthetic heads
    post_cases = [
        ("p544_mixed_b2", (544, 544), 2, "mixed", 11),
        ("p544_dense_b1", (544, 544), 1, "dense", 12),
        ("p544_clustered_b1", (544, 544), 1, "clustered", 13),
        ("p544_sparse_b2", (544, 544), 2, "sparse", 14),
        ("p544_sparse_many_b1", (544, 544), 1, "sparse_many", 15),
        ("p544_empty_b1", (544, 544), 1, "empty", 16),
        ("p96_mixed_b3", (96, 96), 3, "mixed", 17),
        ("p96_dense_b2", (96, 96), 2, "dense", 18),
        ("p96_clustered_b2", (96, 96), 2, "clustered", 19),
        ("p160x128_mixed_b2", (160, 128), 2, "mixed", 20),
        ("p160x128_sparse_b2", (160, 128), 2, "sparse", 21),
        ("p160x128_sparse_many_b2", (160, 128), 2, "sparse_many", 22),
        # adversarial near-ties: scores 1-3 ulps apart straddling the nms_pre cut, scores stepping
        # through conf_thresh, IoUs stepping through the NMS threshold a fraction of an ulp at a time
        ("p544_ties_cut_b2", (544, 544), 2, "ties_cut", 31),
        ("p544_ties_thresh_b2", (544, 544), 2, "ties_thresh", 32),
        ("p544_ties_iou_b1", (544, 544), 1, "ties_iou", 33),
    ]
    for name, size, batch, regime, seed in post_cases:
        pc = post_cfg(size)
        post = reval.OrienMaskYOLOPostProcess(nms_func=functools.partial(rfunc.batched_nms, threshold=0.5), **pc)
        heads = synth.synth_heads(seed, batch, pc["grid_size"], regime=regime)
        with torch.no_grad(), single_thread():
            res = post(heads)
        if regime.startswith("ties"):
            check_tie_fixture(post, heads, regime, res)
        rec = dict(size=np.array(size), batch=np.int64(batch), seed=np.int64(seed), regime=np.array(regime))
        for b, r in enumerate(res):
            rec["bbox%d" % b] = r["bbox"].numpy()
            rec["cls%d" % b] = r["cls"].numpy()
            rec["mask%d" % b] = pack_masks(r["mask"].numpy())
            rec["maskshape%d" % b] = np.array(r["mask"].shape)
        np.savez_compressed(os.path.join(OUT, "post_%s.npz" % name), **rec)
        print(name, [int(r["bbox"].shape[0]) for r in res],
              "%.0f KB" % (os.path.getsize(os.path.join(OUT, "post_%s.npz" % name)) / 1024))

    # ---------------------------------------------------------------- G1/G2/G5: forward (+ end to end)
    mcfg = dict(cfg.orienmask_yolo_coco_544_anchor4_fpn_plus_infer["model"])
    mcfg.pop("type"); mcfg["pretrained"] = None
    net = rmodel.OrienMaskYOLOFPNPlus(**mcfg).eval()
    fwd_cases = [
        ("f96_b2", 1, (96, 96), 2, 21, -22.0, 4.0),
        # head_gain 0.7 / obj_bias -3: logits far from saturation, no two of the 400 best scores equal (with head_gain 4 the
        # sigmoids saturate and hundreds of pairs tie at exactly 1.0: which of them torch.topk keeps is unspecified)
        ("f160x128_b1", 2, (160, 128), 1, 22, -3.0, 0.7),
        ("f544_b1", 3, (544, 544), 1, 23, -16.0, 4.0),           # the bench's weights (saturated heads, some exact score ties)
        # six images: above the batch size at which om_forward switches the stride-1 3x3 layers from Winograd F(2x2,3x3) to
        # F(2x4,3x3), so that path is pinned against the reference's own tensors and END-TO-END detections too
        ("f544_b6", 4, (544, 544), 6, 28, -3.0, 0.7),
        # few pairs pass: no top-k, survivors are emitted in candidate order (the other branch of postprocess.py:107,150)
        ("f544_sparse_b2", 5, (544, 544), 2, 29, -8.03, 0.4),
    ]
    for name, wseed, size, batch, xseed, obj_bias, head_gain in fwd_cases:
        sd = synth.synth_state_dict(wseed, obj_bias=obj_bias, head_gain=head_gain)
        x = synth.synth_image_batch(xseed, batch, size[0], size[1])
        rec = dict(size=np.array(size), batch=np.int64(batch), wseed=np.int64(wseed), xseed=np.int64(xseed),
                   obj_bias=np.float32(obj_bias), head_gain=np.float32(head_gain))
        write_forward_fixture(name, net, reval, rfunc, sd, x, rec, size)


def write_forward_fixture(name, net, reval, rfunc, sd, x, rec, size):
    """Reference forward + reference postprocess of one (weights, input) pair -> tests/golden/fwd_<name>.npz."""
    net.load_state_dict(sd, strict=True)       # proves the 524 keys line up
    with torch.no_grad():
        feats = {}
        x32, x16, x8, x4 = net.backbone(x)
        feats.update(x32=x32, x16=x16, x8=x8, x4=x4)
        out = net(x)
    tensors = dict(bbox32=out[0][0], bbox16=out[1][0], bbox8=out[2][0],
                   oriens=torch.cat([out[0][1], out[1][1], out[2][1]], 1), **feats)
    full = size[0] <= 160
    for k, t in tensors.items():
        assert torch.isfinite(t).all(), k
        s, samp, idx = digest(t)
        rec[k + "_sum"] = s; rec[k + "_samples"] = samp; rec[k + "_idx"] = idx
        rec[k + "_shape"] = np.array(t.shape); rec[k + "_absmax"] = np.float32(t.abs().max().item())
        if full and k in ("bbox32", "bbox16", "bbox8", "oriens"):
            rec[k] = t.numpy()
    # end to end through the reference postprocess
    pc = post_cfg(size)
    post = reval.OrienMaskYOLOPostProcess(nms_func=functools.partial(rfunc.batched_nms, threshold=0.5), **pc)
    with torch.no_grad(), single_thread():
        res = post(out)
    for b, r in enumerate(res):
        rec["bbox_det%d" % b] = r["bbox"].numpy(); rec["cls_det%d" % b] = r["cls"].numpy()
        rec["mask%d" % b] = pack_masks(r["mask"].numpy()); rec["maskshape%d" % b] = np.array(r["mask"].shape)
    np.savez_compressed(os.path.join(OUT, "fwd_%s.npz" % name), **rec)
    print(name, {k: float(rec[k + "_absmax"]) for k in ("x4", "x32", "bbox32", "bbox8", "oriens")},
          [int(r["bbox"].shape[0]) for r in res], "exact score ties among the detections:",
          [int(r["bbox"].shape[0] - np.unique(r["bbox"][:, 4].numpy()).size) for r in res],
          "%.0f KB" % (os.path.getsize(os.path.join(OUT, "fwd_%s.npz" % name)) / 1024))
    return tensors


def stress_golden():
    """G9: heavy-tailed weights and inputs for the split-operand precision mode.  BatchNorm scales
    log-uniform over four decades, running_var down to 1e-6, convolution rows of magnitude 1e-20 and exactly zero, saturated and
    near-zero image regions (orienmask_amd/synth.py: synth_state_dict_stress, synth_image_batch_stress).  Every convolution's
    weights carry a normalisation factor measured HERE through the reference model (pre-BatchNorm output rms 1 on the fixture's
    own input, layer by layer in execution order) and stored in the fixture, so that the 75-layer-deep activations neither
    vanish nor overflow and every machine regenerates identical weights."""
    import torch.nn.functional as F
    from orienmask_amd import arch
    cfg, rmodel, reval, rfunc = import_reference()
    torch.set_num_threads(8)
    mcfg = dict(cfg.orienmask_yolo_coco_544_anchor4_fpn_plus_infer["model"])
    mcfg.pop("type"); mcfg["pretrained"] = None
    net = rmodel.OrienMaskYOLOFPNPlus(**mcfg).eval()
    specs = list(arch.model_convs("OrienMaskYOLOFPNPlus"))
    for name, wseed, size, batch, xseed in (("stress_f160x128_b1", 51, (160, 128), 1, 52), ("stress_f544_b2", 53, (544, 544), 2, 54)):
        x = synth.synth_image_batch_stress(xseed, batch, size[0], size[1])
        net.load_state_dict(synth.synth_state_dict_stress(wseed), strict=True)
        norms = np.ones(len(specs), dtype=np.float32)
        mods = dict(net.named_modules())
        hooks = []
        for li, spec in enumerate(specs):
            m = mods[spec.name + (".conv_block.0" if spec.bn else "")]

            def pre(mod, inp, li=li):
                rows = mod.weight.abs().amax(dim=(1, 2, 3)) > 1e-10             # the 1e-20 and zero rows keep their magnitude
                y = F.conv2d(inp[0], mod.weight[rows], None, mod.stride, mod.padding)
                n = np.float32(1.0 / float(y.double().pow(2).mean().sqrt()))
                norms[li] = n
                mod.weight.data[rows] *= float(n)

            hooks.append(m.register_forward_pre_hook(pre))
        with torch.no_grad():
            net(x)
        for h in hooks:
            h.remove()
        sd = synth.synth_state_dict_stress(wseed, norms)
        rec = dict(size=np.array(size), batch=np.int64(batch), wseed=np.int64(wseed), xseed=np.int64(xseed),
                   obj_bias=np.float32(-3.0), head_gain=np.float32(0.7), stress=np.int64(1), norms=norms)
        tensors = write_forward_fixture(name, net, reval, rfunc, sd, x, rec, size)
        # what the fixture stresses, on the reference's own activations
        with torch.no_grad():
            acts = []
            hs = [m.register_forward_hook(lambda mod, i, o: acts.append(o)) for n_, m in net.named_modules()
                  if m.__class__.__name__ == "LeakyReLU"]
            net(x)
            for h in hs:
                h.remove()
        amax = max(float(a.abs().max()) for a in acts)
        small = float(np.mean([float((a.abs() < 1e-4).float().mean()) for a in acts]))
        print("  %s: %d activation tensors, largest |a| = %.1f, mean fraction of |a| < 1e-4: %.3f, norms %.3g .. %.3g"
              % (name, len(acts), amax, small, norms.min(), norms.max()))
        assert amax < 3000.0, "the stress fixture must stay inside the split representation's range (else it tests the fallback)"


def trained_golden():
    """G10 (task 8): a forward fixture whose weights have the statistics of a converged DarkNet-53 + FPNPlus --
    BatchNorm running statistics matched to their own inputs, heavy-tailed gamma / beta (orienmask_amd/synth.py:
    synth_state_dict_trained) -- so that the split representation's range guard meets per-stage activation maxima in the
    hundreds.  The running statistics are measured HERE through the reference model, layer by layer in execution order on the
    fixture's own input (with a ~20 % mismatch drawn from a seeded generator, as between a dataset and one batch of it), and stored
    in the fixture."""
    from orienmask_amd import arch
    cfg, rmodel, reval, rfunc = import_reference()
    torch.set_num_threads(8)
    mcfg = dict(cfg.orienmask_yolo_coco_544_anchor4_fpn_plus_infer["model"])
    mcfg.pop("type"); mcfg["pretrained"] = None
    net = rmodel.OrienMaskYOLOFPNPlus(**mcfg).eval()
    specs = list(arch.model_convs("OrienMaskYOLOFPNPlus"))
    for name, wseed, size, batch, xseed in (("trained_f544_b2", 71, (544, 544), 2, 72),):
        x = synth.synth_image_batch_stress(xseed, batch, size[0], size[1])       # saturated / near-zero regions and ramps over noise
        net.load_state_dict(synth.synth_state_dict_trained(wseed), strict=True)
        jit = np.random.Generator(np.random.PCG64(wseed + 1000))
        mods = dict(net.named_modules())
        means, vars_, head_norms, hooks = {}, {}, {}, []
        for li, spec in enumerate(specs):
            if spec.bn:
                bn = mods[spec.name + ".conv_block.1"]

                def pre_bn(mod, inp, li=li):
                    y = inp[0].double()
                    m = y.mean(dim=(0, 2, 3)); v = y.var(dim=(0, 2, 3), unbiased=False)
                    c = m.numel()
                    v_run = (v * torch.from_numpy(np.exp(jit.standard_normal(c) * 0.2))).clamp_min(1e-8)
                    m_run = m + v.sqrt() * torch.from_numpy(jit.standard_normal(c) * 0.1)
                    means[li] = m_run.float().numpy(); vars_[li] = v_run.float().numpy()
                    mod.running_mean.copy_(m_run.float()); mod.running_var.copy_(v_run.float())

                hooks.append(bn.register_forward_pre_hook(pre_bn))
            else:
                conv = mods[spec.name]

                def pre_head(mod, inp, li=li):
                    # He-initialised rows on a unit-rms input give logits of rms ~ sqrt(2) x the per-channel gain
                    n = np.float32(1.0 / float(inp[0].double().pow(2).mean().sqrt()))
                    head_norms[li] = n
                    mod.weight.data *= float(n)

                hooks.append(conv.register_forward_pre_hook(pre_head))
        with torch.no_grad():
            net(x)
        for h in hooks:
            h.remove()
        bn_mean = np.concatenate([means[li] for li, sp in enumerate(specs) if sp.bn]).astype(np.float32)
        bn_var = np.concatenate([vars_[li] for li, sp in enumerate(specs) if sp.bn]).astype(np.float32)
        hn = np.array([head_norms[li] for li, sp in enumerate(specs) if not sp.bn], dtype=np.float32)
        sd = synth.synth_state_dict_trained(wseed, (bn_mean, bn_var, hn))
        rec = dict(size=np.array(size), batch=np.int64(batch), wseed=np.int64(wseed), xseed=np.int64(xseed),
                   obj_bias=np.float32(-3.0), head_gain=np.float32(0.7), trained=np.int64(1), bn_mean=bn_mean, bn_var=bn_var,
                   head_norms=hn)
        write_forward_fixture(name, net, reval, rfunc, sd, x, rec, size)
        # what the fixture exercises, on the reference's own activations: per-stage maxima and the bulk
        with torch.no_grad():
            acts = []
            hs = [m.register_forward_hook(lambda mod, i, o, n_=n_: acts.append((n_, o))) for n_, m in net.named_modules()
                  if m.__class__.__name__ == "LeakyReLU"]
            feats = net.backbone(x)
            net(x)
            for h in hs:
                h.remove()
        stage = {}
        for n_, a in acts:
            key = ".".join(n_.split(".")[:2]) if n_.startswith("backbone") else n_.split(".")[0]
            stage[key] = max(stage.get(key, 0.0), float(a.abs().max()))
        amax = max(stage.values())
        rms = float(np.mean([float(a.double().pow(2).mean().sqrt()) for _, a in acts]))
        res_max = [float(f.abs().max()) for f in feats]
        print("  %s: %d activation tensors, per-stage largest |a|: %s; residual-stream outputs x32/x16/x8/x4: %s; mean rms %.2f"
              % (name, len(acts), {k: round(v, 1) for k, v in stage.items()}, [round(v, 1) for v in res_max], rms))
        assert 100.0 < amax < 6000.0 and max(res_max) > 100.0, "trained-like: per-stage maxima in the hundreds, inside the split range"


def check_tie_fixture(post, heads, regime, res):
    """The adversarial fixtures must actually be adversarial: assert the near-tie structure on the reference's own numbers."""
    for b in range(heads[0][0].shape[0]):
        confs = []
        for i in range(post.scales):
            nA, nH, nW = post.num_anchors[i], post.nHs[i], post.nWs[i]
            t = heads[i][0][b].view(nA, -1, nH, nW).permute(0, 2, 3, 1).contiguous()
            _, conf = post.get_boxes(t, nH, nW, post.normalized_anchors[post.anchor_mask[i]], post.grid_x[i], post.grid_y[i])
            confs.append(conf)
        conf = torch.cat(confs, 0)
        passing = np.sort(conf[conf > post.conf_thresh].numpy())[::-1]
        bits = passing.view(np.int32).astype(np.int64)
        if regime == "ties_cut":
            assert passing.size > post.nms_pre
            gap = bits[post.nms_pre - 1] - bits[post.nms_pre]
            assert 1 <= gap <= 3, gap                                  # the cut separates scores 1-3 ulps apart
            lad = bits[passing < 0.5]
            assert (np.diff(lad) < 0).all() and (np.diff(lad) >= -4).all()
            assert int((res[b]["bbox"][:, 4] < 0.5).sum()) > 20      # ladder members are visible in the output
        elif regime == "ties_thresh":
            thr = np.float32(post.conf_thresh)
            near = conf[(conf > 0.00499) & (conf < 0.00501)].numpy()
            d = near.view(np.int32).astype(np.int64) - int(thr.view(np.int32))
            assert (d > 0).sum() >= 30 and (d <= 0).sum() >= 30 and np.abs(d).min() <= 2
            assert res[b]["bbox"].shape[0] < post.nms_post             # nothing is cut away after the threshold
        elif regime == "ties_iou":
            k = res[b]["bbox"].shape[0]
            assert 64 + 10 <= k <= 128 - 10 and k <= post.nms_post, k  # the IoU = 0.5 crossing lies inside the ladder


def yolo_golden():
    """G7: the non-Plus OrienMaskYOLO (model/orienmask_yolo.py) forward on synthetic weights."""
    cfg, rmodel, reval, rfunc = import_reference()
    net = rmodel.OrienMaskYOLO(num_anchors=3, num_classes=80, pretrained=None).eval()
    rec = {}
    for name, wseed, size, batch, xseed in (("y96_b2", 41, (96, 96), 2, 42), ("y128x160_b1", 43, (128, 160), 1, 44)):
        sd = synth.synth_state_dict(wseed, obj_bias=-16.0, head_gain=4.0, model="OrienMaskYOLO")
        net.load_state_dict(sd, strict=True)                       # proves the 506 keys line up
        x = synth.synth_image_batch(xseed, batch, size[0], size[1])
        with torch.no_grad():
            out = net(x)
        rec[name + "_meta"] = np.array([wseed, xseed, batch, size[0], size[1]])
        for k, t in (("bbox32", out[0][0]), ("bbox16", out[1][0]), ("bbox8", out[2][0]),
                     ("oriens", torch.cat([out[0][1], out[1][1], out[2][1]], 1))):
            rec["%s_%s" % (name, k)] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "yolo_fwd.npz"), **rec)
    print("yolo_fwd: %.0f KB" % (os.path.getsize(os.path.join(OUT, "yolo_fwd.npz")) / 1024))


def coco_format_golden():
    """G8: COCOMetrics._recover_shape_bbox / _recover_shape_segm run from the reference (static methods)."""
    cfg, rmodel, reval, rfunc = import_reference()
    from eval.coco_eval import COCOMetrics
    rng = np.random.Generator(np.random.PCG64(61))
    cases = [
        ("plain", (544, 544), dict(height=480, width=640)),
        ("collate", (544, 544), dict(height=427, width=640, collate_pad=[0, 0, 0, 0, 544, 544])),
        ("padded", (320, 352), dict(height=300, width=333, collate_pad=[9, 10, 10, 10, 320, 352])),
        ("both", (160, 192), dict(height=97, width=131, collate_pad=[8, 8, 0, 0, 160, 192], pad=[5, 6, 7, 8, 160, 176])),
        ("flips", (96, 128), dict(height=211, width=150, hflip=True, vflip=True, pad=[3, 0, 0, 4, 96, 128])),
    ]
    rec = {}
    for name, (H, W), info in cases:
        K = 7
        # blobby masks: thresholded smooth noise
        yy, xx = np.mgrid[0:H, 0:W]
        masks = np.zeros((K, H, W), dtype=bool)
        for k in range(K):
            cy, cx = rng.random(2) * [H, W]
            ry, rx = (0.05 + 0.3 * rng.random(2)) * [H, W]
            masks[k] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1) ^ (rng.random((H, W)) < 0.02)
        bbox = np.concatenate([rng.random((K, 2)), rng.random((K, 2)) * 0.5, rng.random((K, 1))], 1).astype(np.float32)
        xywh = COCOMetrics._recover_shape_bbox(torch.from_numpy(bbox[:, :4]), info)
        seg = COCOMetrics._recover_shape_segm(torch.from_numpy(masks), info)
        rec[name + "_hw"] = np.array([H, W]); rec[name + "_info"] = np.array(json_dumps(info))
        rec[name + "_mask"] = pack_masks(masks); rec[name + "_bbox"] = bbox
        rec[name + "_xywh"] = xywh.numpy(); rec[name + "_seg"] = pack_masks(seg.numpy()); rec[name + "_segshape"] = np.array(seg.shape)
    np.savez_compressed(os.path.join(OUT, "coco_format.npz"), **rec)
    print("coco_format: %d cases, %.0f KB" % (len(cases), os.path.getsize(os.path.join(OUT, "coco_format.npz")) / 1024))


def json_dumps(obj):
    import json
    return json.dumps(obj)


def preprocess_golden():
    """G6: FastCOCOTransform (Resize + Normalize) and infer.pad run from the reference's own code.
    data/transform.py needs cv2 / torchvision at import time only (module-level tables and the CPU
    transforms), so they are stubbed; infer.py imports the whole application, so its `pad` function
    is compiled from the file's AST and executed here (nothing of it is stored)."""
    import ast
    import math
    cv2 = types.ModuleType("cv2")
    for i, n in enumerate(("INTER_NEAREST", "INTER_LINEAR", "INTER_AREA", "INTER_CUBIC", "INTER_LANCZOS4")):
        setattr(cv2, n, i)
    sys.modules["cv2"] = cv2
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                 "torchvision.transforms.transforms"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms.transforms"].Lambda = object
    sys.modules["torchvision.transforms.transforms"].Compose = object
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_transform", "/root/reference/data/transform.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    T = mod.FastCOCOTransform
    tree = ast.parse(open("/root/reference/infer.py").read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "pad"][0]
    ns = {"math": math, "F": torch.nn.functional}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "infer.py:pad", "exec"), ns)
    ref_pad = ns["pad"]
    rec = {}
    cases = [("vga", 2, 480, 640, (544, 544)), ("coco", 1, 427, 640, (544, 544)), ("small", 3, 97, 131, (96, 160)),
             ("same", 1, 64, 96, None), ("odd", 1, 333, 500, (300, 451))]
    for k, (name, n, h, w, size) in enumerate(cases):
        img = synth.synth_photo_batch(31 + k, n, h, w)
        x = img.permute(0, 3, 1, 2).contiguous()                      # FastCOCOTransform.__call__, transform.py:459
        if size is not None:
            x = T.Resize(size=size, interpolation="bilinear", align_corners=False)(x)
        x = T.Normalize(mean=(0, 0, 0), std=(255, 255, 255))(x)
        padded, info = ref_pad(x)
        rec[name + "_seed"] = np.int64(31 + k); rec[name + "_shape"] = np.array([n, h, w])
        rec[name + "_size"] = np.array(size if size is not None else (h, w))
        rec[name + "_pad"] = np.array(info); rec[name + "_outshape"] = np.array(padded.shape)
        flat = padded.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 4096).long()
        rec[name + "_idx"] = idx.numpy(); rec[name + "_samples"] = flat[idx].numpy()
        rec[name + "_sum"] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
        if padded.numel() <= 200000:
            rec[name + "_out"] = padded.numpy()
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **rec)
    print("preprocess: %d cases, %.0f KB" % (len(cases), os.path.getsize(os.path.join(OUT, "preprocess.npz")) / 1024))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "preprocess":
        preprocess_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "yolo":
        yolo_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "coco":
        coco_format_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "stress":
        stress_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "trained":
        trained_golden()
    else:
        main()
        yolo_golden()
        coco_format_golden()
        preprocess_golden()
        stress_golden()
        trained_golden()
