"""Per-group time line of conv_wino14.hip from a trace build (tools/build_variant.sh w14trace "-DOM_W14_TRACE=1" conv_wino14):
   gpurun -- 'OM_LIB=ab/w14trace.so python tools/wino14_trace.py'
Stamps per group (shader cycles): consumers  a = group start, b = weight DMA issued, c = last matrix instruction issued,
d = operands landed (then the barrier); producers  a = start, b = input landed, c = transform + V stores issued, d = stores done."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from orienmask_amd import lib as omlib  # noqa: E402
from orienmask_amd.pack import winograd14_weights_split  # noqa: E402

SHAPES = [(68, 128, 256), (34, 256, 512), (136, 128, 256)]


def main():
    omlib.LIB_PATH = os.path.abspath(os.environ["OM_LIB"])
    L = omlib.load()
    raw = ctypes.CDLL(omlib.LIB_PATH)
    dev = torch.device("cuda:0")
    B = 32
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for hw, cin, cout in SHAPES:
        x = torch.randn(B, hw, hw, cin, device=dev)
        w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
        out = torch.empty(B, hw, hw, cout, device=dev)
        hd = torch.zeros(cout, device=dev)
        u14, e14 = winograd14_weights_split(w, cout)
        s14 = torch.pow(torch.tensor(2.0), -e14.float()).to(dev)
        u14 = u14.to(dev)
        trace = torch.zeros(8 * 12 * 64 * 4, dtype=torch.int64, device=dev)
        raw.om_debug_w14_trace(p(trace))
        st = omlib.current_stream_ptr(dev)
        for _ in range(3):
            trace.zero_()
            omlib.check(L.om_conv2d_wino14_split(p(x), B, hw, hw, cin, cin, p(u14), p(s14), p(hd), cout, 1, None, 0, p(out), cout, None, st), "w14")
        torch.cuda.synchronize()
        t = trace.cpu().view(8, 12, 64, 4)
        print("== %dx%d %d->%d" % (hw, hw, cin, cout))
        for blk in range(8):
            if t[blk, 0, 6, 0] == 0:
                continue
            t0 = int(t[blk, 0, 6, 0])
            print(" block %d (first tile; groups 6..23; cycles from group 6's start)" % blk)
            print("  per group: cycles from the group's start (consumer wave 0) at which each wave reached the barrier -- consumers 0-7: operands landed; producers 8-11: work issued")
            for g in range(6, min(24, 6 * (cin // 16))):
                a0 = int(t[blk, 0, g, 0])
                arr = [int(t[blk, w, g, 3 if w < 8 else 2]) - a0 for w in range(12)]
                nxt = int(t[blk, 0, g + 1, 0]) - a0 if g + 1 < 6 * (cin // 16) else 0
                print("  g%2d  %s  | group %5d" % (g, " ".join("%5d" % v for v in arr), nxt))
            ph = [int(v) for v in t[blk, 0, 60]]
            print("   consumer wave 0, first tile: main loop %d cycles, epilogue (+ next tile's weight requests) %d cycles" % (ph[1] - ph[0], ph[2] - ph[1]))
            prev_end = None
            for i in range(8):
                q = [int(v) for v in t[blk, 1, 48 + i]]
                if q[0] == 0:
                    break
                gap = "" if prev_end is None else "  (epilogue end -> this loop's start: %d)" % (q[0] - prev_end)
                print("   tile %d of this workgroup: main loop %6d, epilogue %6d%s" % (i, q[1] - q[0], q[2] - q[1], gap))
                prev_end = q[2]
            pp = [int(v) for v in t[blk, 8, 62]]
            print("   producer wave 8, a later tile: prologue (ticket, chunk 0 landed + transformed, chunk 1 requested) %d cycles" % (pp[1] - pp[0]))
            break


if __name__ == "__main__":
    main()
