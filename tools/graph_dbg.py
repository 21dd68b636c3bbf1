"""Scratch check: hipGraph replay of the forward vs the eager call on a second model instance, fresh inputs each
replay (no workspace clearing in between), then timing.  usage: python tools/graph_dbg.py <B>"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
dev = torch.device('cuda:0')
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
net = OrienMaskYOLOFPNPlus(3, 80).eval(); net.load_state_dict(sd); net = net.to(dev)
net2 = OrienMaskYOLOFPNPlus(3, 80).eval(); net2.load_state_dict(sd); net2 = net2.to(dev)
B = int(sys.argv[1])
x = synth.synth_image_batch(5, B, 544, 544).to(dev)
with torch.no_grad():
    static_in = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): net(static_in)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(static_in)
    torch.cuda.synchronize()
    for seed in (11, 12, 13, 14):
        x2 = synth.synth_image_batch(seed, B, 544, 544).to(dev)
        want2 = [(a.clone(), b.clone()) for a, b in net2(x2)]
        static_in.copy_(x2); g.replay(); torch.cuda.synchronize()
        ok = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(out, want2))
        print("B", B, "replay-after-replay, new input seed", seed, "match", ok, flush=True)
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); print("graph ms", (time.perf_counter() - t0) * 100)
    t0 = time.perf_counter()
    for _ in range(10): net(x)
    torch.cuda.synchronize(); print("eager ms", (time.perf_counter() - t0) * 100)
