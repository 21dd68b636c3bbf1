"""Small-batch sweep of conv_igemm_split.hip: every distinct layer shape of the 544 x 544 forward at batch B (default 1) -- the
stride-1 3x3 layers as the direct convolutions of the latency mode -- with each tile shape and each number of split-K parts
(om_conv2d_split_k) -> one line per shape, best first.  Input for conv_tile_for_split's latency rule and om_model_set_latency_ksplit.
    python tools/latency_tile_sweep.py [B] [only3x3|only1x1]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orienmask_amd import arch, lib as omlib          # noqa: E402
from orienmask_amd.pack import conv_weights_split     # noqa: E402

TILES = [(128, 128), (128, 64), (64, 64), (128, 32)]
PARTS = [1, 2, 3, 4, 6, 8]


def main():
    dev = torch.device("cuda", 0)
    L = omlib.load()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    S = 544
    shapes = {}
    for s in arch.fpnplus_convs():
        if s.name == "backbone.conv1":
            continue
        if (only == "only3x3" and s.ksize != 3) or (only == "only1x1" and s.ksize != 1):
            continue
        div = arch.layer_div(s)                      # of the layer's OUTPUT
        key = (S // div * s.stride, s.cin, s.cout, s.ksize, s.stride)
        shapes.setdefault(key, []).append(s.name)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    total_auto = total_best = 0.0
    for (hw, cin, cout, k, stride), names in shapes.items():
        if cin % 16:
            continue
        x = torch.randn(B, hw, hw, cin, device=dev)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        cpad = (cout + 31) // 32 * 32
        ws, e = conv_weights_split(w, cpad)
        wd = ws.to(dev)
        sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
        hp = torch.zeros(cpad, device=dev)
        ho = hw // stride
        out = torch.empty(B, ho, ho, cpad, device=dev)

        def timed(run):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / 20

        def auto():
            omlib.check(L.om_conv2d_split(p(x), B, hw, hw, cin, cin, p(wd), p(sp), p(hp), cout, k, stride, 1, None, 0,
                                          p(out), cpad, 0, 1, 0, 0, None, omlib.current_stream_ptr(dev)), "conv")
        t_auto = timed(auto)
        res = []
        for bm, bn in TILES:
            if cpad % bn:
                continue
            for parts in PARTS:
                def run():
                    omlib.check(L.om_conv2d_split_k(p(x), B, hw, hw, cin, cin, p(wd), p(sp), p(hp), cout, k, stride, 1, None, 0,
                                                    p(out), cpad, 0, 1, bm, bn, parts, None, omlib.current_stream_ptr(dev)), "conv")
                res.append((timed(run), bm, bn, parts))
                if bn != 64:
                    break                      # no split-K form of that shape
        res.sort()
        total_auto += t_auto * len(names)
        total_best += res[0][0] * len(names)
        M = B * ho * ho
        print("hw=%3d cin=%4d cout=%4d k=%d s=%d M=%6d x%2d  auto %.4f | " % (hw, cin, cout, k, stride, M, len(names), t_auto) +
              "  ".join("%dx%d/%d %.4f" % (bm, bn, parts, ms) for ms, bm, bn, parts in res[:6]), flush=True)
    print("sum over the forward's layers: chooser %.3f ms, best per shape %.3f ms" % (total_auto, total_best))


if __name__ == "__main__":
    main()
