"""Wall-time split of COCOFormatter.to_coco_format on one batch of 8 images x 100 masks (544^2 -> 480x640):
   gpurun -- 'python tools/coco_time.py'"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from orienmask_amd import synth  # noqa: E402
from orienmask_amd.coco_format import COCOFormatter  # noqa: E402
from orienmask_amd.eval import OrienMaskYOLOPostProcess  # noqa: E402

dev = torch.device("cuda:0")
pc = bench.post_config(544, 544)
heads = synth.synth_heads(5, 8, pc["grid_size"], regime="dense")
post = OrienMaskYOLOPostProcess(device=dev, **pc)
dets = post(tuple((b.to(dev), o.to(dev)) for b, o in heads))
infos = [dict(id=i, height=480, width=640, collate_pad=[0, 0, 0, 0, 544, 544]) for i in range(len(dets))]
fmt = COCOFormatter(list(range(1, 81)))
fmt.to_coco_format(infos[:2], dets[:2])
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    res = fmt.to_coco_format(infos, dets)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("masks", len(res["segm"]), "ms/image %.3f" % (dt / len(dets) * 1e3), "bytes of strings", sum(len(s["segmentation"]["counts"]) for s in res["segm"]))
if os.environ.get("OM_COCO_PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    fmt.to_coco_format(infos, dets)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
