"""Time every distinct 1x1 / stride-2 layer shape of the bs=32, 544x544 forward with each tile shape of conv_igemm_split.hip
(om_conv2d_split's per-call tile_bm x tile_bn) -> one line per (shape, tile), best first.  Input for the tile chooser in conv_igemm_split.hip."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orienmask_amd import arch, lib as omlib          # noqa: E402
from orienmask_amd.pack import conv_weights_split     # noqa: E402

TILES = [(256, 128), (128, 128), (128, 64), (64, 64), (128, 32)]


def main():
    dev = torch.device("cuda", 0)
    L = omlib.load()
    B, S = 32, 544
    shapes = {}
    for s in arch.fpnplus_convs():
        if s.name == "backbone.conv1" or (s.ksize == 3 and s.stride == 1):
            continue
        div = arch.layer_div(s)                      # of the layer's OUTPUT
        key = (S // div * s.stride, s.cin, s.cout, s.ksize, s.stride)
        shapes.setdefault(key, []).append(s.name)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for (hw, cin, cout, k, stride), names in shapes.items():
        x = torch.randn(B, hw, hw, cin, device=dev)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        cpad = (cout + 31) // 32 * 32
        ws, e = conv_weights_split(w, cpad)
        wd = ws.to(dev)
        sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
        hp = torch.zeros(cpad, device=dev)
        ho = hw // stride
        out = torch.empty(B, ho, ho, cpad, device=dev)
        res = []
        for bm, bn in TILES:
            if cpad % bn:
                continue
            def run():
                omlib.check(L.om_conv2d_split(p(x), B, hw, hw, cin, cin, p(wd), p(sp), p(hp), cout, k, stride, 1, None, 0,
                                              p(out), cpad, 0, 1, bm, bn, None, omlib.current_stream_ptr(dev)), "conv")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run()
            b.record()
            torch.cuda.synchronize()
            res.append((a.elapsed_time(b) / 10, bm, bn))
        res.sort()
        M = B * ho * ho
        print("hw=%3d cin=%4d cout=%4d k=%d s=%d M=%7d x%2d  " % (hw, cin, cout, k, stride, M, len(names)) +
              "  ".join("%dx%d %.3f" % (bm, bn, ms) for ms, bm, bn in res), flush=True)


if __name__ == "__main__":
    main()
