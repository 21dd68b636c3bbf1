"""Micro-benchmark of single fused convolutions through the C ABI (om_conv2d) on an MI355X.

    python tools/conv_bench.py [--batch 32] [--iters 20] [--shapes neck4.1,conv4.conv.1,...]

Shapes are named after the layers of the 544x544 forward they come from.
"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from orienmask_amd import lib as omlib  # noqa: E402

#          name              H    cin  cout k  s  res
SHAPES = {
    "conv2.0":      (544, 32, 64, 3, 2, False),
    "conv2.1.c1":   (272, 32, 64, 3, 1, True),
    "conv3.0":      (272, 64, 128, 3, 2, False),
    "conv3.c0":     (136, 128, 64, 1, 1, False),
    "conv3.c1":     (136, 64, 128, 3, 1, True),
    "neck4.1":      (136, 128, 256, 3, 1, False),
    "neck4.0":      (136, 256, 128, 1, 1, False),
    "conv4.c0":     (68, 256, 128, 1, 1, False),
    "conv4.c1":     (68, 128, 256, 3, 1, True),
    "neck8.1":      (68, 128, 256, 3, 1, False),
    "conv5.c0":     (34, 512, 256, 1, 1, False),
    "conv5.c1":     (34, 256, 512, 3, 1, True),
    "conv6.c0":     (17, 1024, 512, 1, 1, False),
    "conv6.c1":     (17, 512, 1024, 3, 1, True),
    "head8.1":      (68, 256, 255, 1, 1, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--wino", action="store_true", help="run the 3x3 stride-1 shapes through Winograd F(2x2,3x3)")
    args = ap.parse_args()
    L = omlib.load()
    dev = torch.device("cuda:0")
    B = args.batch
    for name in args.shapes.split(","):
        H, cin, cout, k, s, res = SHAPES[name]
        Ho = H // s
        cpad = (cout + 31) // 32 * 32
        x = torch.randn(B, H, H, cin, device=dev)
        w = torch.randn(cpad, k * k * cin, device=dev) * 0.05
        sc = torch.ones(cpad, device=dev); sh = torch.zeros(cpad, device=dev)
        r = torch.randn(B, Ho, Ho, cout, device=dev) if res else None
        out = torch.empty(B, Ho, Ho, cout, device=dev)
        st = omlib.current_stream_ptr(dev)

        wino = args.wino and k == 3 and s == 1
        if wino:
            from orienmask_amd.pack import winograd_weights
            cpad = (cout + 63) // 64 * 64
            w = winograd_weights(torch.randn(cout, cin, 3, 3) * 0.05, cpad).to(dev)
            sc = torch.ones(cpad, device=dev); sh = torch.zeros(cpad, device=dev)
            scratch = torch.empty(L.om_conv2d_winograd_scratch_bytes(B, H, H, cin), dtype=torch.uint8, device=dev)

        def run():
            if wino:
                rc = L.om_conv2d_winograd(ctypes.c_void_p(x.data_ptr()), B, H, H, cin, cin, ctypes.c_void_p(w.data_ptr()),
                                          ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(sh.data_ptr()), cout, 1,
                                          ctypes.c_void_p(r.data_ptr()) if res else None, cout if res else 0,
                                          ctypes.c_void_p(out.data_ptr()), cout, ctypes.c_void_p(scratch.data_ptr()),
                                          scratch.numel(), st)
                omlib.check(rc, "om_conv2d_winograd")
                return
            rc = L.om_conv2d(ctypes.c_void_p(x.data_ptr()), B, H, H, cin, cin, ctypes.c_void_p(w.data_ptr()),
                             ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(sh.data_ptr()), cout, k, s, 1,
                             ctypes.c_void_p(r.data_ptr()) if res else None, cout if res else 0,
                             ctypes.c_void_p(out.data_ptr()), cout, st)
            omlib.check(rc, "om_conv2d")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        flops = 2.0 * B * Ho * Ho * cout * cin * k * k
        byts = 4.0 * (B * H * H * cin + B * Ho * Ho * cout * (2 if res else 1) + cout * cin * k * k)
        print("%-12s H=%3d %4d->%4d k%d s%d res=%d  %8.3f ms  %7.2f TF  %7.1f GB/s" %
              (name, H, cin, cout, k, s, int(res), ms, flops / ms / 1e9, byts / ms / 1e6))


if __name__ == "__main__":
    main()
