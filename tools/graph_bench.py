import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
from orienmask_amd.eval import OrienMaskYOLOPostProcess
from orienmask_amd.graph import GraphedPipeline
from bench import post_config
dev = torch.device('cuda:0')
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
net = OrienMaskYOLOFPNPlus(3, 80).eval(); net.load_state_dict(sd); net = net.to(dev)
post = OrienMaskYOLOPostProcess(device=dev, **post_config(544, 544))
for B in [int(v) for v in sys.argv[1:]] or [1, 4, 32]:
    x = synth.synth_image_batch(5, B, 544, 544).to(dev)
    pipe = GraphedPipeline(net, post, x)
    for name, fn in (("eager", lambda: post(net(x))), ("graph", lambda: pipe(x))):
        with torch.no_grad():
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 30 if B < 32 else 10
            for _ in range(n): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("B=%2d %-6s %7.3f ms/step %8.1f img/s" % (B, name, dt * 1e3, B / dt))
