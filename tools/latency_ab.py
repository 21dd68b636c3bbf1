#!/usr/bin/env python
"""bs = 1 (and a few more) latency of forward + postprocess through the captured hipGraph, one at a time, for several settings of
the latency mode:  python tools/latency_ab.py [--bs 1 2 4] [--ksplit 1 2 4 8] [--layers]
(random-init weights of the flagship configuration, 544 x 544; what bench.py's `small_batches` times for one setting)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, nargs="+", default=[1])
    ap.add_argument("--ksplit", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--lib", default=None, help="another build of the library (ab/NAME.so)")
    ap.add_argument("--fuse", type=int, nargs="+", default=[1], help="1: eval.launch_step (decode + select beside the orientation branch), 0: two calls on one stream")
    ap.add_argument("--cells", type=int, default=None, help="the switch of the latency mode in 1/32-scale cells per batch (default: the model's)")
    ap.add_argument("--layers", action="store_true", help="per-layer kernel times of the last setting (HIP events, eager)")
    args = ap.parse_args()
    if args.lib:
        from orienmask_amd import lib as _omlib
        _omlib.LIB_PATH = os.path.abspath(args.lib)
    import bench
    from orienmask_amd import synth
    from orienmask_amd.eval import OrienMaskYOLOPostProcess
    from orienmask_amd.graph import GraphedPipeline
    from orienmask_amd.model import OrienMaskYOLOFPNPlus
    dev = torch.device("cuda:0")
    net = OrienMaskYOLOFPNPlus(3, 80).eval().set_precision("f32_split")
    net.load_state_dict(synth.synth_state_dict(bench.WEIGHT_SEED, obj_bias=bench.OBJ_BIAS, head_gain=bench.HEAD_GAIN), strict=True)
    net = net.to(dev)
    post = OrienMaskYOLOPostProcess(device=dev, **bench.post_config(544, 544))
    for bsz in args.bs:
        xs = [torch.rand(bsz, 3, 544, 544, generator=torch.Generator().manual_seed(s)).to(dev) for s in (1, 2)]

        def time_loop(fn, n):
            for _ in range(5):
                fn(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        with torch.no_grad():
            net.set_latency_mode(False)
            gp = GraphedPipeline(net, post, xs[0])
            print("bs=%d default                 %.3f ms" % (bsz, time_loop(lambda i: gp(xs[i & 1]), args.iters)), flush=True)
            del gp
            for ks in args.ksplit:
                for br in args.fuse:
                    net.set_latency_mode(True, cells=args.cells, ksplit=ks)
                    gp = GraphedPipeline(net, post, xs[0], fuse_step=bool(br))
                    ms = [time_loop(lambda i: gp(xs[i & 1]), args.iters) for _ in range(3)]
                    print("bs=%d latency mode, ksplit %d fused step %d   %s ms" % (bsz, ks, br, " ".join("%.3f" % m for m in ms)), flush=True)
                    del gp
            if args.layers:      # per-layer table of this batch size in latency mode (bench.py's events), to stderr of that run
                import subprocess
                subprocess.call([sys.executable, "bench.py", "--batch", str(bsz), "--steps", "10", "--warmup", "3", "--latency-mode",
                                 "--no-small-batch", "--layers"])
            net.set_latency_mode(False)


if __name__ == "__main__":
    main()
