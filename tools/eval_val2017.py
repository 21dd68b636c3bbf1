"""val2017 mask / box AP of a reference checkpoint on the HIP path -- the harness BASELINE.json's "mask AP within 0.1 of reference
on val2017" needs (not runnable offline -- no checkpoint, no COCO, no pycocotools in this image).

    python tools/eval_val2017.py --ckpt checkpoints/OrienMaskAnchor4FPNPlus/orienmask_yolo.pth --coco-root /data/coco
    python tools/eval_val2017.py --synthetic 12 --out /tmp/val_dry          # this container / the GPU box: everything up to the
                                                                             # json files pycocotools would be handed

It is /root/reference/test.py:18-29 -> trainer/builder.py:43-58 -> trainer/tester.py:26-52 with this repo's pieces:
  build_tester      orienmask_amd.builder.build_tester: the MODEL config comes from the checkpoint's own train config,
                    weights load strictly, the postprocess is config/base.py:219-236 (`orienmask_yolo_coco_544_anchor4_postprocess`)
  test loader       COCODataset + transform_val_544 (/root/reference/data/dataset.py:41-100, config/base.py:154-162, 179-188):
                    images of coco/list/coco_val.txt (or every image of instances_val2017.json), RGB float32, warped to 544 x 544
                    (`pad_needed=False`), / 255, batches of 16 (config/config_test.py:9-12); `batch_info` = {'id','height','width'}.
                    The warp runs on the GPU (orienmask_amd.transform.FastCOCOTransform = F.interpolate bilinear,
                    align_corners=False); the reference's test loader uses cv2.resize INTER_LINEAR on the host -- the same
                    half-pixel-centre formula in float32, equal up to rounding (the infer.py path uses exactly ours).
  Convert Format    orienmask_amd.coco_format.COCOFormatter (boxes / masks recovered and RLE-packed on the device)
  coco_eval         /root/reference/eval/coco_eval.py:80-101 verbatim in behaviour: bbox_prediction.json / segm_prediction.json ->
                    pycocotools COCO.loadRes -> COCOeval('bbox' | 'segm') when pycocotools is importable.
Published reference numbers (/root/reference/assets/val2017_test_result.log:1-6,40-42): bbox AP 0.385, segm AP 0.345; the run
passes when both are within --tol (default 0.001) of them, exit code 0; 3 when pycocotools / data are missing (json files still
written), 1 when the APs disagree.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

ANCHORS_YOLOV4 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
ANCHORS_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]
# /root/reference/data/dataset.py:42-48
CAT2LABEL = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34, 35, 36,
             37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 67, 70,
             72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90]
REFERENCE_AP = {"bbox": 0.385, "segm": 0.345}          # assets/val2017_test_result.log:6,42
METRIC_KEYS = ["AP", "AP50", "AP75", "APS", "APM", "APL", "AR1", "AR10", "AR100", "ARS", "ARM", "ARL"]

TEST_CONFIG = dict(      # config/config_test.py:4-15 without the loader (built below) and the model (from the checkpoint)
    postprocess=dict(type="OrienMaskYOLOPostProcess", grid_size=[[17, 17], [34, 34], [68, 68]], image_size=[544, 544],
                     anchors=ANCHORS_YOLOV4, anchor_mask=ANCHORS_MASK, num_classes=80, conf_thresh=0.005,
                     nms=dict(type="batched_nms", threshold=0.5), nms_pre=400, nms_post=100, orien_thresh=0.3),
    gt_file="coco/annotations/instances_val2017.json")
DEFAULT_MODEL = dict(type="OrienMaskYOLOFPNPlus", num_anchors=3, num_classes=80, pretrained=None, freeze_backbone=False,
                     backbone_batchnorm_eval=False)


class Val2017Loader:
    """COCODataset(with_info=True) + transform_val_544 + collate, image side only (the evaluation never reads the annotations)."""

    def __init__(self, coco_root, batch_size=16, size=(544, 544), limit=None, device="cuda"):
        from PIL import Image  # noqa: F401  (fail here, not in the middle of the run)
        from orienmask_amd.transform import FastCOCOTransform as T
        self.root = coco_root
        self.batch_size = batch_size
        self.device = device
        self.tf = T([T.Resize(size), T.Normalize((0, 0, 0), (255, 255, 255))])
        self.items = self._list(limit)

    def _list(self, limit):
        lst = os.path.join(self.root, "list", "coco_val.txt")
        ann = os.path.join(self.root, "annotations", "orienmask_coco_val.json")
        items = []
        if os.path.exists(lst) and os.path.exists(ann):           # the reference's own list + its converted annotation file
            anno = json.load(open(ann))
            for line in open(lst):
                name = line.strip().split(",")[0]
                if name:
                    items.append((name, anno[name]["image_id"]))
        else:                                                     # plain COCO layout: every image of instances_val2017.json
            gt = json.load(open(os.path.join(self.root, "annotations", "instances_val2017.json")))
            items = [(im["file_name"], im["id"]) for im in sorted(gt["images"], key=lambda im: im["id"])]
        items = items[:limit] if limit else items
        from orienmask_amd.dist import shard_range, world_info
        rank, world = world_info()                       # one process per GPU: this rank's contiguous slice of the image list
        start, stop = shard_range(len(items), rank, world)
        self.total = len(items)
        return items[start:stop]

    def __len__(self):
        return (len(self.items) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        import numpy as np
        from PIL import Image
        for s in range(0, len(self.items), self.batch_size):
            imgs, infos = [], []
            for name, image_id in self.items[s:s + self.batch_size]:
                rgb = np.asarray(Image.open(os.path.join(self.root, "val2017", name)).convert("RGB"), dtype=np.float32)
                x = self.tf(torch.from_numpy(rgb).to(self.device).unsqueeze(0))       # [1,3,544,544], / 255
                imgs.append(x)
                infos.append(dict(id=int(image_id), height=int(rgb.shape[0]), width=int(rgb.shape[1])))
            yield torch.cat(imgs, dim=0), None, infos


def synthetic_checkpoint():
    """A reference-format checkpoint object ({'state_dict', 'config'}: trainer/base.py:143-152) with seeded random weights."""
    from orienmask_amd import synth
    return {"epoch": 0, "state_dict": synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0), "config": {"model": dict(DEFAULT_MODEL)}}


def coco_eval(gt_file, out_dir, with_mask=True):
    """coco_eval.py:80-101: returns {'bbox': stats[12], 'segm': stats[12]}, or None when pycocotools is absent."""
    try:
        from pycocotools.coco import COCO
        from pycocotools.cocoeval import COCOeval
    except ImportError:
        return None
    gt = COCO(gt_file)
    out = {}
    for kind in (("bbox", "segm") if with_mask else ("bbox",)):
        pred = gt.loadRes(os.path.join(out_dir, "%s_prediction.json" % kind))
        ev = COCOeval(gt, pred, iouType=kind)
        ev.evaluate(); ev.accumulate(); ev.summarize()
        out[kind] = [float(v) for v in ev.stats]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", "-w", default=None, help="reference checkpoint (.pth with 'state_dict' and 'config')")
    ap.add_argument("--coco-root", default="coco", help="directory holding val2017/, annotations/ (and list/coco_val.txt)")
    ap.add_argument("--out", default="val2017_eval", help="where bbox_prediction.json / segm_prediction.json / ap.json go")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--limit", type=int, default=None, help="first N images only (smoke run; the AP check is skipped)")
    ap.add_argument("--precision", default=None, choices=("f32", "f32_split", "f16"))
    ap.add_argument("--in-flight", type=int, default=1, help="batches in flight (1 = the reference's loop and timers)")
    ap.add_argument("--tol", type=float, default=0.001)
    ap.add_argument("--synthetic", type=int, default=0,
                    help="no data: N synthetic images through a seeded random checkpoint, up to the json files (no AP)")
    args = ap.parse_args()
    from orienmask_amd.builder import build_tester
    from orienmask_amd.coco_format import COCOFormatter
    from orienmask_amd.tester import SyntheticLoader
    if not torch.cuda.is_available():
        raise SystemExit("eval_val2017.py needs an MI355X (the product has no CPU path)")
    # one process per GPU (python -m torch.distributed.run --nproc-per-node N tools/eval_val2017.py ...): every rank evaluates its
    # slice of the image list, rank 0 merges and scores -- /root/reference/trainer/trainer.py:175-181,201-205
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    os.makedirs(args.out, exist_ok=True)
    if args.synthetic:
        ckpt = synthetic_checkpoint()
        loader = SyntheticLoader(args.synthetic, args.batch, seed=17, device=dev)
    else:
        if not args.ckpt:
            raise SystemExit("--ckpt is required (or --synthetic N)")
        ckpt = torch.load(args.ckpt, map_location="cpu", weights_only=False)
        if not (isinstance(ckpt, dict) and isinstance(ckpt.get("config"), dict) and "model" in ckpt["config"]):
            sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt      # a bare state_dict: infer.py:81-83
            ckpt = {"state_dict": sd, "config": {"model": dict(DEFAULT_MODEL)}}
        loader = Val2017Loader(args.coco_root, args.batch, limit=args.limit, device=dev)
    fmt = COCOFormatter(CAT2LABEL, with_mask=True)

    def on_batch(batch_info, detections):
        return fmt.to_coco_format(batch_info, detections)          # {'bbox': [...], 'segm': [...]} of this batch

    tester = build_tester(TEST_CONFIG, ckpt, loader, device=dev, on_batch=on_batch)
    if args.precision:
        tester.model.set_precision(args.precision)
    t0 = time.perf_counter()
    try:
        stats, merged = tester.test_and_gather(verbose=rank == 0, in_flight=args.in_flight)
    except BaseException:
        # a rank that fails before the final all_gather_object would leave the others waiting for the collective's time-out:
        # say why and leave at once with a non-zero status (torch.distributed.run then ends the other ranks)
        import traceback
        traceback.print_exc()
        sys.stderr.write("eval_val2017: rank %d failed during evaluation; aborting the job\n" % rank)
        sys.stderr.flush()
        os._exit(4)
    wall = time.perf_counter() - t0
    results = {"bbox": [r for part in merged for r in part["bbox"]], "segm": [r for part in merged for r in part["segm"]]}
    if rank != 0:                 # rank 0 writes and scores
        if world > 1:
            torch.distributed.destroy_process_group()
        return 0
    for kind in ("bbox", "segm"):
        with open(os.path.join(args.out, "%s_prediction.json" % kind), "w") as f:
            json.dump(results[kind], f)
    n_img = args.synthetic if args.synthetic else loader.total
    summary = dict(images=n_img, ranks=world, detections=len(results["bbox"]), wall_s=round(wall, 2), precision=tester.model.precision,
                   speed=stats, reference_ap=REFERENCE_AP)
    rc = 3
    gt_file = os.path.join(args.coco_root, "annotations", "instances_val2017.json")
    if not args.synthetic and os.path.exists(gt_file):
        ev = coco_eval(gt_file, args.out)
        if ev is not None:
            summary["eval"] = {k: dict(zip(METRIC_KEYS, v)) for k, v in ev.items()}
            if args.limit:
                rc = 0
                summary["verdict"] = "partial run (--limit): APs printed, not compared"
            else:
                diffs = {k: abs(round(ev[k][0], 3) - REFERENCE_AP[k]) for k in REFERENCE_AP}
                ok = all(d <= args.tol + 1e-9 for d in diffs.values())
                summary["verdict"] = "PASS" if ok else "FAIL"
                summary["ap_abs_diff"] = diffs
                rc = 0 if ok else 1
        else:
            summary["verdict"] = "pycocotools is not importable: json files written, AP not computed"
    else:
        summary["verdict"] = "synthetic dry run: json files written, nothing to score" if args.synthetic else \
            "no %s: json files written, AP not computed" % gt_file
    with open(os.path.join(args.out, "ap.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "speed"}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
