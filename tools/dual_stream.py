"""Experiment: one batch of B images as S independent sub-batches on S HIP streams vs one stream.
usage: python tools/dual_stream.py <B> <S>"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
dev = torch.device('cuda:0')
B, S = int(sys.argv[1]), int(sys.argv[2])
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
nets = []
for _ in range(S):
    n = OrienMaskYOLOFPNPlus(3, 80).eval(); n.load_state_dict(sd); nets.append(n.to(dev))
x = synth.synth_image_batch(5, B, 544, 544).to(dev)
parts = list(x.chunk(S))
streams = [torch.cuda.Stream() for _ in range(S)]
def step_multi():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for n, p, s in zip(nets, parts, streams):
        with torch.cuda.stream(s):
            n(p)
    for s in streams: cur.wait_stream(s)
def step_single():
    nets[0](x)
with torch.no_grad():
    for name, fn in (("single", step_single), ("multi%d" % S, step_multi), ("single", step_single), ("multi%d" % S, step_multi)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("B=%d %-8s %8.3f ms/step %8.1f img/s" % (B, name, dt * 1e3, B / dt), flush=True)
