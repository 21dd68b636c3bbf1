"""Soak test of InFlightPipeline: N batches through two-in-flight forward + postprocess, every result compared bit for bit
with the one-at-a-time result of the same input (4 distinct inputs, cycled).  usage: python tools/soak_in_flight.py [f32|f16] [N] [B]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
from orienmask_amd.eval import OrienMaskYOLOPostProcess
from orienmask_amd.pipeline import InFlightPipeline
from bench import post_config
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda:0')
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
net = OrienMaskYOLOFPNPlus(3, 80).eval(); net.load_state_dict(sd); net = net.to(dev).set_precision(prec)
post = OrienMaskYOLOPostProcess(device=dev, **post_config(544, 544))
xs = [synth.synth_image_batch(700 + i, B, 544, 544).to(dev) for i in range(4)]
with torch.no_grad():
    want = [[{k: v.clone() for k, v in d.items()} for d in post(net(x))] for x in xs]
pipe = InFlightPipeline(net, post, depth=2)
bad = 0
t0 = time.perf_counter()
for i, dets in enumerate(pipe.map(xs[i % 4] for i in range(N))):
    for g, w in zip(dets, want[i % 4]):
        if not (torch.equal(g["bbox"], w["bbox"]) and torch.equal(g["cls"], w["cls"]) and torch.equal(g["mask"], w["mask"])):
            bad += 1
            print("batch %d: mismatch (mask bytes differing: %d)" % (i, int((g["mask"] != w["mask"]).sum()) if g["mask"].shape == w["mask"].shape else -1), flush=True)
torch.cuda.synchronize()
print("%s: %d batches of %d through two in flight, %d images differ from the one-at-a-time result; %.1f s" % (prec, N, B, bad, time.perf_counter() - t0))
