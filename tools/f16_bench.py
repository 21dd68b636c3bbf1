"""fp16-activation configuration: forward-only timing and per-layer profile.  usage: python tools/f16_bench.py [B]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from orienmask_amd import arch, synth
from orienmask_amd.model import OrienMaskYOLOFPNPlus
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = synth.synth_state_dict(3, obj_bias=-16.0, head_gain=4.0)
net = OrienMaskYOLOFPNPlus(3, 80).eval(); net.load_state_dict(sd); net = net.to(dev).set_precision("f16")
x = synth.synth_image_batch(5, B, 544, 544).to(dev)
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("B=%d f16 forward %8.3f ms/step %8.1f img/s" % (B, dt * 1e3, B / dt), flush=True)
    net.profile_enable(True)
    for _ in range(5): net(x)
    rows, nf = net.profile_read()
    net.profile_enable(False)
kern = dict(net.layer_kernels(B, 544, 544))
specs = {s.name: s for s in arch.model_convs("OrienMaskYOLOFPNPlus", 3, 80)}
agg = {}
for name, ms, pre in rows:
    ms /= nf
    w = arch.layer_work(specs[name], B, 544, 544)
    print("%-28s %7.3f ms %8.1f TF %8.1f GB/s(f16)  %s" % (name, ms, w["flops"] / ms / 1e9, w["bytes"] / 2 / ms / 1e6, kern[name]))
    a = agg.setdefault(kern[name], [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += w["flops"]; a[3] += w["bytes"] / 2
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-36s %3d launches %7.3f ms (%4.1f%%) %8.1f TF %8.1f GB/s" % (k, a[0], a[1], 100 * a[1] / tot, a[2] / a[1] / 1e9, a[3] / a[1] / 1e6))
print("sum of layers %.3f ms" % tot)
