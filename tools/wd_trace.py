"""Per-group time line of the dual-role fused 3x3 kernel (trace build of conv_wino14d.hip: -DOM_WD_TRACE=1):
   tools/build_variant.sh wdtrace "-DOM_WD_TRACE=1" conv_wino14d;  gpurun -- 'OM_LIB=ab/wdtrace.so python tools/wd_trace.py [shape index]'"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from orienmask_amd import lib as omlib  # noqa: E402
from orienmask_amd.pack import winograd14_weights_split  # noqa: E402

SHAPES = [(272, 32, 64), (136, 64, 128), (68, 128, 256), (34, 256, 512), (17, 512, 1024), (136, 128, 256)]


def main():
    omlib.LIB_PATH = os.path.abspath(os.environ["OM_LIB"])
    L = omlib.load()
    dev = torch.device("cuda:0")
    B = 32
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    hw, cin, cout = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 5]
    x = torch.randn(B, hw, hw, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    out = torch.empty(B, hw, hw, cout, device=dev)
    hd = torch.zeros(cout, device=dev)
    u14, e14 = winograd14_weights_split(w, cout)
    s14 = torch.pow(torch.tensor(2.0), -e14.float()).to(dev)
    u14 = u14.to(dev)
    st = omlib.current_stream_ptr(dev)
    trace = torch.zeros(8 * 4 * 512 * 6, dtype=torch.int64, device=dev)
    L.om_debug_wd_trace.argtypes = [ctypes.c_void_p]
    L.om_debug_wd_trace(p(trace))
    L.om_set_wino14_variant(1)
    for _ in range(3):
        omlib.check(L.om_conv2d_wino14_split(p(x), B, hw, hw, cin, cin, p(u14), p(s14), p(hd), cout, 1, None, 0, p(out), cout, None, st), "w14")
    torch.cuda.synchronize()
    steps = trace.cpu()[8 * 4 * 512 * 4:].view(8, 4, 512, 2)
    t = trace.cpu()[:8 * 4 * 512 * 4].view(8, 4, 512, 4)
    nch = cin // 16
    gpt = 6 * nch
    print("%dx%d %d->%d: %d groups per tile" % (hw, hw, cin, cout, gpt))
    for wg in (0, 3):
        for wave in (0, 3):
            tt = t[wg, wave]
            n = int((tt[:, 0] > 0).sum())
            print("workgroup %d wave %d: %d groups traced" % (wg, wave, n))
            # per position q: issue phase (t1 - t0), end wait (t2 - t1), barrier (next t0 - t2); tiles 1.. (steady state)
            acc = collections.defaultdict(list)
            for g in range(gpt, min(n - 1, 500)):
                q = g % 6
                first = (g % gpt) < 6
                key = ("first chunk " if first else "steady      ") + "q=%d" % q
                if (g + 1) % gpt == 0:
                    continue        # the group in front of an epilogue: its "barrier" time is the epilogue
                st = steps[wg, wave, g]
                acc[key].append((int(tt[g, 1] - tt[g, 0]), int(tt[g, 2] - tt[g, 1]), int(tt[g + 1, 0] - tt[g, 2]),
                                 int(st[0] - tt[g, 0]), int(st[1] - st[0]), int(tt[g, 1] - st[1])))
            for k in sorted(acc):
                v = acc[k]
                m = [sum(a[i] for a in v) / len(v) for i in range(6)]
                print("  %s  issue %6.0f (kernel rows %4.0f %4.0f %4.0f)  end wait %6.0f  barrier %6.0f  total %6.0f   (n=%d)" % (
                    k, m[0], m[3], m[4], m[5], m[1], m[2], sum(m[:3]), len(v)))
            epi = []
            for g in range(gpt - 1, min(n - 1, 500), gpt):
                if tt[g, 3] > 0:
                    epi.append((int(tt[g, 3] - tt[g, 2]), int(tt[g + 1, 0] - tt[g, 3])))
            if epi:
                print("  epilogue %6.0f cycles, to the next tile's first group %6.0f  (n=%d)" % (
                    sum(a[0] for a in epi) / len(epi), sum(a[1] for a in epi) / len(epi), len(epi)))
            tiles = [int(tt[g + gpt, 0] - tt[g, 0]) for g in range(gpt, min(n - gpt - 1, 500 - gpt), gpt)]
            if tiles:
                print("  tile to tile: %s" % tiles[:12])


if __name__ == "__main__":
    main()
