"""Soak of the latency mode's fence-free split-K hand-off: N replays of the captured step (one 544 x 544 image, alternating between
two images) while a second stream keeps the chip busy with batches of six; every replay's detections bit for bit against the
eager two-call result of the same image.   python tools/soak_latency.py [N=5000]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                            # noqa: E402
from orienmask_amd import synth                         # noqa: E402
from orienmask_amd.eval import OrienMaskYOLOPostProcess  # noqa: E402
from orienmask_amd.graph import GraphedPipeline          # noqa: E402
from orienmask_amd.model import OrienMaskYOLOFPNPlus     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(bench.WEIGHT_SEED, obj_bias=bench.OBJ_BIAS, head_gain=bench.HEAD_GAIN)


def model(prec):
    m = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision(prec)
    m.load_state_dict(sd, strict=True)
    return m.to(dev)


net, busy = model("f32_split").set_latency_mode(True), model("f16")
post = OrienMaskYOLOPostProcess(device=dev, **bench.post_config(544, 544))
xs = [synth.synth_image_batch(50 + i, 1, 544, 544).to(dev) for i in range(2)]
y = synth.synth_image_batch(60, 6, 544, 544).to(dev)
with torch.no_grad():
    want = [[{k: v.clone() for k, v in d.items()} for d in post(net(x))] for x in xs]
    gp = GraphedPipeline(net, post, xs[0])
    s1 = torch.cuda.Stream(dev)
    bad = 0
    t0 = time.time()
    for i in range(N):
        if i % 8 == 0:
            with torch.cuda.stream(s1):
                busy(y)
        got = gp(xs[i & 1])
        for g, w in zip(got, want[i & 1]):
            if not (torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"])):
                bad += 1
    torch.cuda.synchronize()
print("latency-mode soak: %d replays beside an fp16 neighbour stream, %d differ from the eager result; %.1f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
