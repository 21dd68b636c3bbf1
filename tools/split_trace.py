"""Per-tile time line of conv_igemm_split_wide_kernel from a trace build (tools/build_variant.sh splittrace "-DOM_SPLIT_TRACE=1" conv_igemm_split):
   gpurun -- 'OM_LIB=ab/splittrace.so python tools/split_trace.py'
Per traced tile (shader cycles): ticket + prologue (first stage landed), k loop, of which waiting at the stage barriers, epilogue."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from orienmask_amd import lib as omlib  # noqa: E402
from orienmask_amd.pack import conv_weights_split  # noqa: E402

SHAPES = [(136, 256, 128, 1, 1), (68, 256, 128, 1, 1), (34, 512, 256, 1, 1), (136, 128, 256, 3, 2), (34, 256, 512, 3, 2)]


def main():
    omlib.LIB_PATH = os.path.abspath(os.environ["OM_LIB"])
    L = omlib.load()
    raw = ctypes.CDLL(omlib.LIB_PATH)
    dev = torch.device("cuda:0")
    B = 32
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for hw, cin, cout, k, stride in SHAPES:
        x = torch.randn(B, hw, hw, cin, device=dev)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        ws, e = conv_weights_split(w, cout)
        wd = ws.to(dev)
        sp = torch.pow(torch.tensor(2.0), -e.float()).to(dev)
        hp = torch.zeros(cout, device=dev)
        ho = hw // stride
        out = torch.empty(B, ho, ho, cout, device=dev)
        trace = torch.zeros(16 * 16 * 8, dtype=torch.int64, device=dev)
        raw.om_debug_split_trace(p(trace))
        st = omlib.current_stream_ptr(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            trace.zero_()
            a.record()
            omlib.check(L.om_conv2d_split(p(x), B, hw, hw, cin, cin, p(wd), p(sp), p(hp), cout, k, stride, 1, None, 0, p(out), cout, 0, 1, 0, 0,
                                          None, st), "conv")
            b.record()
        torch.cuda.synchronize()
        t = trace.cpu().view(16, 16, 8)
        M = B * ho * ho
        print("== %dx%d %d->%d k%d s%d  M=%d  %.3f ms, %d stages per tile" % (hw, hw, cin, cout, k, stride, M, a.elapsed_time(b), k * k * cin // 32))
        for blk in (0, 5, 11):
            for i in range(16):
                q = [int(v) for v in t[blk, i]]
                if q[0] == 0:
                    break
                print("  wg %2d tile %5d: ticket+prologue %6d  k loop %6d (waiting %6d)  epilogue %6d   total %6d" % (
                    blk, q[5], q[1] - q[0], q[2] - q[1], q[4], q[3] - q[2], q[3] - q[0]))


if __name__ == "__main__":
    main()
