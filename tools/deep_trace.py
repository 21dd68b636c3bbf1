"""Time line of the deep-ring (latency-mode) implicit GEMM from a trace build:
   tools/build_variant.sh deeptrace "-DOM_SPLIT_TRACE=1" conv_igemm_split;  gpurun -- 'python tools/deep_trace.py'
Per workgroup (s_memtime deltas in shader cycles -> us at 2.4 GHz): start -> first stage landed -> k loop done -> arrival counted -> parts summed -> stored,
as percentiles over the launch's workgroups, separately for the parts that only publish and the last arrivals."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from orienmask_amd import lib as omlib  # noqa: E402
from orienmask_amd.pack import conv_weights_split  # noqa: E402

SHAPES = [(17, 512, 1024, 3, 1, 64, 64, 1), (34, 256, 512, 3, 1, 64, 64, 1), (68, 128, 256, 3, 1, 128, 64, 1), (17, 1024, 512, 1, 1, 64, 64, 0),
          (34, 512, 256, 1, 1, 64, 64, 0), (68, 256, 128, 1, 1, 64, 64, 0)]


def main():
    omlib.LIB_PATH = os.path.abspath(os.environ.get("OM_LIB", "ab/deeptrace.so"))
    L = omlib.load()
    raw = ctypes.CDLL(omlib.LIB_PATH)
    dev = torch.device("cuda:0")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    tick_us = 1.0 / 2400.0      # s_memtime deltas are shader cycles (~2.4 GHz); its absolute value differs between XCDs
    for hw, cin, cout, k, stride, bm, bn, use_res in SHAPES:
        x = torch.randn(1, hw, hw, cin, device=dev)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        ws, e = conv_weights_split(w, cout)
        wd = ws.to(dev)
        sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
        hp = torch.zeros(cout, device=dev)
        out = torch.empty(1, hw, hw, cout, device=dev)
        res = torch.randn(1, hw, hw, cout, device=dev) if use_res else None
        trace = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
        raw.om_debug_split_trace(p(trace))
        st = omlib.current_stream_ptr(dev)
        flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            trace.zero_()
            if it == 2 and os.environ.get("COLD"):
                flush.fill_(1)
            a.record()
            omlib.check(L.om_conv2d_split_k(p(x), 1, hw, hw, cin, cin, p(wd), p(sp), p(hp), cout, k, stride, 1, p(res) if use_res else None,
                                            cout if use_res else 0, p(out), cout, 0, 1, bm, bn, 8, None, st), "conv")
            b.record()
        torch.cuda.synchronize()
        t = trace.cpu().view(512, 8)[:, :6]
        t = t[t[:, 0] > 0].double()
        last = t[:, 5] > 0
        print("== %dx%d %d->%d k%d tile %dx%d res %d: %.1f us by events (launch included), %d workgroups (%d store a tile)" % (
            hw, hw, cin, cout, k, bm, bn, use_res, a.elapsed_time(b) * 1e3, len(t), int(last.sum())))

        def pct(v):
            v = v.sort().values
            return "%5.1f / %5.1f / %5.1f" % tuple(float(v[int(q * (len(v) - 1))]) * tick_us for q in (0.1, 0.5, 0.9))
        print("   first stage landed            ", pct(t[:, 1] - t[:, 0]))
        print("   k loop                        ", pct(t[:, 2] - t[:, 1]))
        print("   publish + arrival             ", pct(t[:, 3] - t[:, 2]))
        if last.any():
            tl = t[last]
            print("   last arrival: parts summed    ", pct(tl[:, 4] - tl[:, 3]))
            print("   last arrival: epilogue stored ", pct(tl[:, 5] - tl[:, 4]))
            print("   last arrival: start to stored ", pct(tl[:, 5] - tl[:, 0]))


if __name__ == "__main__":
    main()
