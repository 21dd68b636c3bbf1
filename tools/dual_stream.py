"""Experiment: one batch of B images as S independent sub-batches on S HIP streams vs one stream, forward only and
forward + postprocess.  usage: python tools/dual_stream.py <B> <S> [f16]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
from orienmask_amd.eval import OrienMaskYOLOPostProcess
from bench import post_config
dev = torch.device('cuda:0')
B, S = int(sys.argv[1]), int(sys.argv[2])
prec = "f16" if len(sys.argv) > 3 and sys.argv[3] == "f16" else "f32"
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
nets, posts = [], []
for _ in range(S):
    n = OrienMaskYOLOFPNPlus(3, 80).eval(); n.load_state_dict(sd); nets.append(n.to(dev).set_precision(prec))
    posts.append(OrienMaskYOLOPostProcess(device=dev, **post_config(544, 544)))
x = synth.synth_image_batch(5, B, 544, 544).to(dev)
parts = list(x.chunk(S))
streams = [torch.cuda.Stream() for _ in range(S)]
def fwd_multi():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for n, p, s in zip(nets, parts, streams):
        with torch.cuda.stream(s):
            n(p)
    for s in streams: cur.wait_stream(s)
def fwd_single():
    nets[0](x)
def e2e_single():
    return posts[0](nets[0](x))
def e2e_multi():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    outs = []
    for n, po, p, s in zip(nets, posts, parts, streams):
        with torch.cuda.stream(s):
            outs.append(po.launch(n(p)))
    for s in streams: cur.wait_stream(s)
    return [d for po, o in zip(posts, outs) for d in po.collect(o)]
with torch.no_grad():
    for name, fn in (("fwd single", fwd_single), ("fwd multi%d" % S, fwd_multi), ("e2e single", e2e_single), ("e2e multi%d" % S, e2e_multi),
                     ("fwd single", fwd_single), ("fwd multi%d" % S, fwd_multi), ("e2e single", e2e_single), ("e2e multi%d" % S, e2e_multi)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("%s B=%d %-11s %8.3f ms/step %8.1f img/s" % (prec, B, name, dt * 1e3, B / dt), flush=True)
