"""Experiment: S whole batches of B images in flight on S HIP streams (one model instance and workspace each) vs one
stream; forward only.  usage: python tools/pipelined_steps.py <B> <S> [f16]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
dev = torch.device('cuda:0')
B, S = int(sys.argv[1]), int(sys.argv[2])
prec = "f16" if len(sys.argv) > 3 and sys.argv[3] == "f16" else "f32"
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
nets = []
for _ in range(S):
    n = OrienMaskYOLOFPNPlus(3, 80).eval(); n.load_state_dict(sd); nets.append(n.to(dev).set_precision(prec))
x = synth.synth_image_batch(5, B, 544, 544).to(dev)
streams = [torch.cuda.Stream() for _ in range(S)]
def single(k):
    for _ in range(k): nets[0](x)
def multi(k):
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for i in range(k):
        with torch.cuda.stream(streams[i % S]):
            nets[i % S](x)
    for s in streams: cur.wait_stream(s)
with torch.no_grad():
    for name, fn in (("single", single), ("in flight %d" % S, multi)) * 2:
        fn(4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(20)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("%s B=%d %-12s %8.3f ms/batch %8.1f img/s" % (prec, B, name, dt * 1e3, B / dt), flush=True)
