"""Two batches in flight, the second stream started a fraction of a step behind the first: do complementary layers overlap better?"""
import sys, time, itertools, torch
sys.path.insert(0, '/root/repo')
import bench
from orienmask_amd import synth
from orienmask_amd.eval import OrienMaskYOLOPostProcess
from orienmask_amd.model import OrienMaskYOLOFPNPlus
from orienmask_amd.pipeline import InFlightPipeline
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32_split'
net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision(prec)
net.load_state_dict(synth.synth_state_dict(bench.WEIGHT_SEED, obj_bias=bench.OBJ_BIAS, head_gain=bench.HEAD_GAIN), strict=True)
net = net.to(dev)
post = OrienMaskYOLOPostProcess(device=dev, **bench.post_config(544, 544))
xs = [synth.synth_image_batch(1000 + i, 32, 544, 544).to(dev) for i in range(2)]
n = 60
for rep in range(2):
    for delay_ms in (0.0, 4.0, 8.0, 12.0, 16.0, 22.0):
        pipe = InFlightPipeline(net, post, depth=2)
        with torch.no_grad():
            for _ in pipe.map(itertools.islice(itertools.cycle(xs), 6)):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it = itertools.islice(itertools.cycle(xs), n)
            pipe.submit(next(it))
            time.sleep(delay_ms * 1e-3)
            for _ in pipe.map(it):
                pass
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("%s stagger %5.1f ms  %8.1f images/s" % (prec, delay_ms, 32 * n / dt), flush=True)
        del pipe
